"""Choosing WHERE in HBM a long-lived batch lives.

Second finding (profiles/r02/experiments/placement_spacing.txt, placement_search.txt,
arena_distance_sweep.txt): what makes a placement slow is that the arrays a kernel streams
concurrently lie CLOSE to each other in physical memory.  Arrays allocated one after the other (544 MiB
apart for the headline batch), or carved out of one allocation at any distance up to 2.5 GiB, run the
operator in 0.43-0.44 ms; the same operator on arrays that were allocated 6 or more arrays apart (>= 3 GiB)
runs in 0.385-0.395 ms, reproducibly.  choose_spread() therefore allocates a pool of arrays in one go, takes
arrays that lie `spacing` allocations apart, probes a few such index sets and frees the rest of the pool.

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


def choose_spread(allocate_array, n_arrays, probe, spacing=11, extra=4, max_candidates=8, device=None):
    """allocate_array() -> one freshly allocated device array of the batch (called (n_arrays-1)*(spacing+1)+1+extra
    times in a row; the whole pool is alive at once, everything but the chosen arrays is freed at the end);
    probe(tuple of n_arrays arrays) -> milliseconds of a representative kernel on them, in the kernel's argument
    order (values in the arrays do not matter to the time; fill the chosen ones afterwards).
    Candidates are the index sets (b, b+s, b+2s, ...) for s in (spacing, spacing+1) and the bases b that fit,
    at most max_candidates of them, plus -- for the record -- the first n_arrays arrays of the pool, i.e. what
    allocating the batch in one go would have given.
    Returns (tuple of the chosen arrays, report dict)."""
    if n_arrays < 1 or spacing < 1:
        raise ValueError("n_arrays and spacing must be >= 1")
    size = (n_arrays - 1) * (spacing + 1) + 1 + max(0, extra)
    pool = [allocate_array() for _ in range(size)]
    cands = []
    for s in (spacing, spacing + 1):
        for b in range(0, size - (n_arrays - 1) * s):
            cands.append(tuple(b + k * s for k in range(n_arrays)))
    # spread the candidates over bases and both spacings
    step = max(1, len(cands) // max(1, max_candidates))
    cands = cands[::step][:max_candidates]
    adjacent = tuple(range(n_arrays))
    ms_adjacent = float(probe(tuple(pool[i] for i in adjacent))) if size >= n_arrays else None
    ms = [float(probe(tuple(pool[i] for i in c))) for c in cands]
    best = min(range(len(cands)), key=lambda i: ms[i])
    if ms_adjacent is not None and ms_adjacent < ms[best]:
        chosen_idx, chosen_ms = adjacent, ms_adjacent
    else:
        chosen_idx, chosen_ms = cands[best], ms[best]
    chosen = tuple(pool[i] for i in chosen_idx)
    del pool
    if torch.cuda.is_available():
        torch.cuda.empty_cache()
    return chosen, {"method": "arrays %d allocations apart, pool of %d" % (spacing, size), "candidates": [list(c) for c in cands],
                    "probe_ms": [round(t, 4) for t in ms], "allocated_in_one_go_ms": None if ms_adjacent is None else round(ms_adjacent, 4),
                    "chosen": list(chosen_idx), "chosen_ms": round(chosen_ms, 4)}


def choose_search(allocate_array, n_arrays, probe, pool_size=24, random_sets=24, max_probes=160, seed=5, good_enough_ms=None, device=None):
    """Third form (profiles/r02/experiments/placement_search.txt): which COMBINATION of arrays a kernel streams
    decides its time (no single array and no pair predicts it), candidates spread bimodally (0.40 / 0.435 ms for
    the headline batch), and in some processes none of a handful of structured candidates is a fast one.  So:
    a pool of `pool_size` arrays allocated in one go; probe the first n_arrays ("allocated in one go", for the
    record), the sets that lie pool_size // n_arrays allocations apart and `random_sets` random index sets; then
    coordinate descent from the best -- replace one array of the set at a time by every other array of the pool,
    keep what is faster -- until `max_probes` probes are spent, a full sweep brings nothing or the time is at or
    below `good_enough_ms`.  Everything but the chosen arrays is freed.
    probe(tuple of n_arrays arrays) -> milliseconds, in the kernel's argument order (values do not matter).
    Returns (tuple of the chosen arrays, report dict)."""
    import random

    if n_arrays < 1 or pool_size < n_arrays:
        raise ValueError("pool_size must be >= n_arrays >= 1")
    pool = [allocate_array() for _ in range(pool_size)]
    seen = {}

    def timed(idx):
        idx = tuple(idx)
        if idx not in seen:
            seen[idx] = float(probe(tuple(pool[i] for i in idx)))
        return seen[idx]

    adjacent = tuple(range(n_arrays))
    ms_adjacent = timed(adjacent)
    step = pool_size // n_arrays
    for b in range(step):
        timed(tuple(b + k * step for k in range(n_arrays)))
    rng = random.Random(seed)
    for _ in range(random_sets):
        if len(seen) >= max_probes:
            break
        timed(tuple(rng.sample(range(pool_size), n_arrays)))
    best = min(seen, key=seen.get)
    first_phase = len(seen)
    improved = True
    while improved and len(seen) < max_probes and not (good_enough_ms is not None and seen[best] <= good_enough_ms):
        improved = False
        for pos in list(range(n_arrays // 2, n_arrays)) + list(range(n_arrays // 2)):  # outputs first: the stores are the slower side
            cur = list(best)
            for i in range(pool_size):
                if i in cur or len(seen) >= max_probes:
                    continue
                c = list(cur)
                c[pos] = i
                if timed(c) < seen[best]:
                    best = tuple(c)
                    improved = True
            if good_enough_ms is not None and seen[best] <= good_enough_ms:
                break
    confirm = float(probe(tuple(pool[i] for i in best)))
    chosen = tuple(pool[i] for i in best)
    times = sorted(seen.values())
    del pool
    if torch.cuda.is_available():
        torch.cuda.empty_cache()
    return chosen, {"method": "search over index sets of a pool of %d arrays: %d structured/random sets, then coordinate descent" % (pool_size, first_phase),
                    "probes": len(seen), "allocated_in_one_go_ms": round(ms_adjacent, 4), "probe_ms_min_median_max": [round(times[0], 4), round(times[len(times) // 2], 4), round(times[-1], 4)],
                    "chosen": list(best), "chosen_ms": round(seen[best], 4), "chosen_reprobed_ms": round(confirm, 4)}


def choose_search_rounds(allocate_array, n_arrays, probe, rounds=3, pool_size=48, **search_args):
    """choose_search() over `rounds` DIFFERENT pools, the fastest chosen set kept.  Whether a pool holds a fast set at all
    is a property of the pool (profiles/r02/experiments/placement_pools.txt; in the round-end run of round 2 two of five
    bench.py processes found nothing below 0.407 ms in their one pool of 48 where the others found 0.382).  A freed pool's
    memory is what the next allocations get back, so between rounds the freed part is re-occupied by ballast arrays (and
    the losing sets stay allocated) until the last round is over: peak = rounds * pool_size arrays.
    Returns (tuple of the chosen arrays, report of the winning round + "rounds_chosen_ms")."""
    best = None
    hold = []
    per_round = []
    for r in range(max(1, rounds)):
        chosen, report = choose_search(allocate_array, n_arrays, probe, pool_size=pool_size, seed=5 + r, **search_args)
        per_round.append(report["chosen_reprobed_ms"])
        if best is None or report["chosen_reprobed_ms"] < best[1]["chosen_reprobed_ms"]:
            if best is not None:
                hold.append(best[0])
            best = (chosen, report)
        else:
            hold.append(chosen)
        del chosen
        if r + 1 < rounds:
            hold.append([allocate_array() for _ in range(pool_size - n_arrays)])
    del hold
    if torch.cuda.is_available():
        torch.cuda.empty_cache()
    report = dict(best[1])
    report["rounds_chosen_ms"] = per_round
    report["method"] = report["method"] + "; best of %d pools" % len(per_round)
    return best[0], report

