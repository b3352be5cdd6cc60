"""Choosing WHERE in HBM a long-lived batch lives.

Measured on MI355X (profiles/r02/experiments/alloc_variance.txt, arena_variance.txt, time_series.txt):
the time of a streaming kernel over the same arrays is stable to 0.3 % for as long as the arrays stay
where they are, but changes by 3-9 % when they are freed and allocated again -- even at the same
virtual addresses.  The physical pages an allocation lands on decide how evenly its streams spread over
the memory channels, and neither the caller nor the library can ask for particular pages.  What a caller
CAN do, for a batch that is allocated once and computed on many times (the usual life of a model-level
batch in HBM), is to allocate a few candidates, time a probe on each, keep the fastest and free the
rest.  That is all this module does; it computes nothing and changes no result.
"""
import torch

# sizes (MiB) of the spacer allocations put between candidates, so that successive candidates land on
# different physical pages; odd sizes on purpose
SPACERS_MIB = (0, 515, 2050, 1031, 4099, 259, 3075, 131)


def choose_placement(allocate, probe, tries=6, device=None, spacers_mib=SPACERS_MIB):
    """allocate() -> any object holding freshly allocated device tensors (called `tries` times, all
    candidates alive at once: size the tries to the memory you can spare);
    probe(candidate) -> milliseconds of a representative kernel on it (smaller is better).
    Returns (the chosen candidate, report dict).  Losing candidates and spacers are freed and the
    caching allocator is emptied, the winner stays where it is."""
    if tries < 1:
        raise ValueError("tries must be >= 1")
    held, spacers, ms = [], [], []
    for i in range(tries):
        mib = spacers_mib[i % len(spacers_mib)]
        if mib:
            spacers.append(torch.empty(mib << 20, dtype=torch.uint8, device=device))
        cand = allocate()
        held.append(cand)
        ms.append(float(probe(cand)))
    best = min(range(tries), key=lambda i: ms[i])
    chosen = held[best]
    del held, spacers, cand
    if torch.cuda.is_available():
        torch.cuda.empty_cache()
    return chosen, {"tries": tries, "probe_ms": [round(t, 4) for t in ms], "chosen": best}
