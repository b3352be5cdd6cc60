"""mi_fieldcalc_amd -- MI355X (gfx950) implementation of the mi-fieldcalc hot
path: the elementwise derived-variable operators and the 5-point-stencil
operators of ``miutil::fieldcalc`` (reference: src/mi_fieldcalc/FieldCalculations.h).

This Python layer is plumbing over the C ABI in ``include/mifc.h``: it picks
pointers out of numpy arrays (host memory, the legacy calling convention) or
PyTorch CUDA tensors (fields resident in HBM), forwards to ``libmifc.so`` and
hands back ``(result, fDefined)``.  Operator names, argument order and failure
behaviour follow the reference: where the C++ function returns ``false`` the
wrapper returns ``None`` (as the reference's pybind11 layer does,
python/py_mi_fieldcalc.cc:92-93).

There is no CPU compute path here.  Without the HIP library the import fails,
without a GPU ``Context()`` raises.
"""
import ctypes

import numpy as np

from . import _capi

ALL_DEFINED, NONE_DEFINED, SOME_DEFINED = 0, 1, 2  # miutil::ValuesDefined, FieldDefined.h:41
UNDEF = np.float32(1.0e35)  # miutil::UNDEF, FieldDefined.cc:34
MEM_HOST, MEM_DEVICE = 0, 1

__all__ = ["Context", "SlabPlan", "Graph", "PreparedCalls", "ALL_DEFINED", "NONE_DEFINED", "SOME_DEFINED", "UNDEF", "classify"]


def classify(n_undefined, n):
    """miutil::checkDefined(size_t, size_t), FieldDefined.cc:62-70."""
    return _capi.lib().mifc_classify(int(n_undefined), int(n))


def _is_torch(x):
    return type(x).__module__.startswith("torch")


class _Arg:
    """Address + keep-alive of one field argument.

    levels=True accepts a (nlev, ny, nx) device tensor whose levels are padded
    (stride(0) >= ny*nx, every level itself contiguous): ``lstride`` is that
    level stride in floats.  output=True refuses host arrays that would have to
    be converted (the caller's array would never be written)."""

    __slots__ = ("addr", "keep", "device", "shape", "dev_index", "lstride")

    def __init__(self, x, allow_none=False, levels=False, output=False):
        self.dev_index, self.lstride = None, None
        if x is None:
            if not allow_none:
                raise ValueError("missing field argument")
            self.addr, self.keep, self.device, self.shape = None, None, None, None
            return
        if _is_torch(x):
            import torch

            if x.dtype != torch.float32:
                raise ValueError("device fields must be float32 tensors")
            if not x.is_cuda:
                raise ValueError("torch tensors must live on the GPU; pass numpy arrays for host memory")
            if levels and x.dim() == 3 and not x.is_contiguous():
                nlev, ny, nx = x.shape
                if x.stride(2) != 1 or x.stride(1) != nx or x.stride(0) < ny * nx or x.stride(0) % 4 != 0:
                    raise ValueError("a level batch must be (nlev, ny, nx) with contiguous levels and a level stride that is a multiple of 4")
            elif not x.is_contiguous():
                raise ValueError("device fields must be contiguous float32 tensors")
            self.addr, self.keep, self.device, self.shape = x.data_ptr(), x, True, tuple(x.shape)
            self.dev_index = x.device.index
            if levels and x.dim() == 3:
                self.lstride = int(x.stride(0)) if x.shape[0] > 1 else int(x.shape[1] * x.shape[2])
        else:
            if output and not (isinstance(x, np.ndarray) and x.dtype == np.float32 and x.flags["C_CONTIGUOUS"]):
                raise ValueError("an output array must be a C-contiguous float32 numpy array (or a CUDA tensor)")
            a = np.ascontiguousarray(x, dtype=np.float32)  # forcecast, like py_mi_fieldcalc.cc:40
            self.addr, self.keep, self.device, self.shape = a.ctypes.data, a, False, a.shape
            if levels and a.ndim == 3:
                self.lstride = int(a.shape[1] * a.shape[2])


def _memkind(args, ctx_device=None):
    present = [a for a in args if a.addr is not None]
    kinds = {a.device for a in present}
    if len(kinds) != 1:
        raise ValueError("all fields of one call must be either numpy (host) or CUDA tensors (device)")
    if ctx_device is not None:
        for a in present:
            if a.device and a.dev_index is not None and a.dev_index != ctx_device:
                raise ValueError("a tensor lives on cuda:%d but this Context was created on device %d" % (a.dev_index, ctx_device))
    return MEM_DEVICE if kinds.pop() else MEM_HOST


def _same_shape(args, shape):
    return all(a.addr is None or tuple(a.shape) == tuple(shape) for a in args)


def _empty_like(ref, shape=None):
    if _is_torch(ref):
        import torch

        return torch.empty(shape or ref.shape, dtype=torch.float32, device=ref.device)
    return np.empty(shape or np.shape(ref), dtype=np.float32)


class Context:
    """One mifc_ctx: a HIP device, a stream, staging scratch.  Not thread-safe;
    create one per thread (the reference is re-entrant)."""

    def __init__(self, device=0, stream=None):
        self._recording = None  # prepare(): the C calls made while it runs
        self._lib = _capi.lib()
        self._ctx = self._lib.mifc_create(int(device))
        if not self._ctx:
            raise RuntimeError(
                "mifc_create(%d) failed: no usable HIP device (visible devices: %d). "
                "The operators run on the GPU only; there is no CPU fallback." % (device, self._lib.mifc_device_count())
            )
        self.device = device
        if stream is not None:
            self.set_stream(stream)

    def close(self):
        if getattr(self, "_ctx", None):
            self._lib.mifc_destroy(self._ctx)
            self._ctx = None

    __del__ = close

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    # ------------------------------------------------------------------ misc
    def last_error(self):
        s = self._lib.mifc_last_error(self._ctx)
        return s.decode() if s else ""

    def set_stream(self, stream):
        """stream: raw hipStream_t address (0 = HIP's default stream), a
        torch.cuda.Stream, or None to go back to the context's own stream."""
        if stream is None:
            self._lib.mifc_use_own_stream(self._ctx)
            return
        if hasattr(stream, "cuda_stream"):
            stream = stream.cuda_stream
        self._lib.mifc_set_stream(self._ctx, ctypes.c_void_p(int(stream)))

    def use_torch_stream(self):
        import torch

        self.set_stream(torch.cuda.current_stream(self.device).cuda_stream)

    def reload_env(self):
        """Re-read the MIFC_* tuning / diagnostic variables (they are read when a context is created)."""
        self._lib.mifc_reload_env(self._ctx)

    def synchronize(self):
        if not self._lib.mifc_synchronize(self._ctx):
            raise RuntimeError(self.last_error())

    def _measure_entry(self, name):
        fn = getattr(self._lib, name, None)
        if fn is None:
            raise RuntimeError("%s exists in the measurement build of the library only (tools/ select it through MIFC_LIB_PATH)" % name)
        return fn

    def timing_begin(self):
        """Measurement build only: bracket every kernel launch of the following calls with HIP events."""
        if not self._measure_entry("mifc_timing_begin")(self._ctx):
            raise RuntimeError(self.last_error())

    def timing_end_ms(self):
        """Summed kernel time (ms) of the calls since timing_begin(); -1 if unavailable."""
        return float(self._measure_entry("mifc_timing_end_ms")(self._ctx))

    def hold_field(self, host_array):
        """Uploads a constant host field (map ratios, Coriolis parameter) once;
        host-pointer calls that are handed the same array then skip its upload.
        The array must stay alive and unchanged until release_field()."""
        a = _Arg(host_array)
        if a.device:
            raise ValueError("hold_field is for host arrays")
        if a.keep is not host_array:
            raise ValueError("hold_field needs a C-contiguous float32 array (a converted copy would have another address)")
        self._held = getattr(self, "_held", {})
        self._held[a.addr] = a.keep
        if not self._lib.mifc_hold_field(self._ctx, a.addr, int(a.keep.size)):
            raise RuntimeError(self.last_error())
        return a.keep

    def release_field(self, host_array):
        a = _Arg(host_array)
        self._lib.mifc_release_field(self._ctx, a.addr)
        getattr(self, "_held", {}).pop(a.addr, None)

    # ------------------------------------------------------------ call helper
    def _bind_stream(self, memkind):
        """Fields resident on the device come from PyTorch: run on torch's
        current stream so that the kernels are ordered after whatever produced
        the inputs (and before whatever consumes the outputs).  Host-pointer
        calls use the context's own stream."""
        if getattr(self, "_frozen_stream", False):
            return  # a graph capture is open (Graph): the recorded calls go to the capture stream
        if memkind == MEM_DEVICE:
            self.use_torch_stream()
        else:
            self.set_stream(None)

    def _call(self, name, args):
        # numpy arrays among the arguments are host tables (flags, per-level coefficients): passed by address
        cargs = [a.ctypes.data if isinstance(a, np.ndarray) else a for a in args]
        fn = getattr(self._lib, name)
        if self._recording is not None:
            self._recording.append((name, fn, cargs, [a for a in args if isinstance(a, np.ndarray)]))
        rc = fn(self._ctx, *cargs)
        if not rc and self.last_error():
            err = self.last_error()
            raise RuntimeError("%s: %s" % (name, err))
        return rc

    def prepare(self, fn):
        """Runs fn() -- asynchronous *_enqueue wrapper calls on device tensors -- once and returns a PreparedCalls whose launch()
        repeats exactly the C calls it made (same addresses, same scalars, the host tables kept alive) without the wrapper's
        shape checks, array conversions and stream lookup: ~3 us per call instead of ~30.  The tensors must stay alive and in
        place; the context must be on the stream the launches are meant for (use_torch_stream)."""
        self._recording = []
        try:
            fn()
        finally:
            rec, self._recording = self._recording, None
        return PreparedCalls(self, rec)

    @staticmethod
    def _nxny(a):
        ny, nx = a.shape[-2], a.shape[-1]
        return nx, ny

    def _single(self, name, fields, scalars, outs, fdefined, undef, n_out=1, out_like=None, lead=(), pre=(), tail=()):
        """fields: input arrays (None allowed), scalars: list placed between
        inputs and outputs in reference order, outs: preallocated or None.
        lead: arguments before (nx, ny) (fieldOPER*'s compute), pre: scalars
        before the fields, tail: arguments between the outputs and the flag."""
        fa = [_Arg(f, allow_none=True) for f in fields]
        ref = next(f for f in fields if f is not None)
        ra = _Arg(ref)
        if len(ra.shape) != 2:
            return None  # like the reference's binding (py_mi_fieldcalc.cc:82-83): not a 2-D field
        nx, ny = self._nxny(ra)
        outs = list(outs)
        for k in range(n_out):
            if outs[k] is None:
                outs[k] = _empty_like(ref)
        oa = [_Arg(o, output=True) for o in outs]
        # every field of the call has the shape of the first one; the reference's binding returns None
        # otherwise (same_dims, py_mi_fieldcalc.cc:82-83) -- here it also keeps the kernels inside the buffers
        if not _same_shape(fa + oa, ra.shape):
            return None
        mk = _memkind(fa + oa, self.device)
        self._bind_stream(mk)
        fd = ctypes.c_int(int(fdefined))
        args = (list(lead) + [nx, ny] + list(pre) + [a.addr for a in fa] + list(scalars) + [a.addr for a in oa] + list(tail)
                + [ctypes.addressof(fd), float(undef), mk])
        if not self._call(name, args):
            return None
        res = [o if _is_torch(o) else a.keep for o, a in zip(outs, oa)]
        return (res[0] if n_out == 1 else tuple(res)), fd.value

    # ---------------------------------------------------- elementwise operators
    def vectorabs(self, u, v, fdefined=SOME_DEFINED, undef=UNDEF, out=None):
        """ff = sqrt(u*u+v*v); FieldCalculations.cc:1819."""
        return self._single("mifc_vectorabs", [u, v], [], [out], fdefined, undef)

    def pleveltemp(self, tinp, p, unit, compute, fdefined=SOME_DEFINED, undef=UNDEF, out=None):
        return self._single("mifc_pleveltemp", [tinp], [float(p), unit.encode(), int(compute)], [out], fdefined, undef)

    def hleveltemp(self, tinp, ps, alevel, blevel, unit, compute, fdefined=SOME_DEFINED, undef=UNDEF, out=None):
        return self._single(
            "mifc_hleveltemp", [tinp, ps], [float(alevel), float(blevel), unit.encode(), int(compute)], [out], fdefined, undef
        )

    def aleveltemp(self, tinp, p, unit, compute, fdefined=SOME_DEFINED, undef=UNDEF, out=None):
        return self._single("mifc_aleveltemp", [tinp, p], [unit.encode(), int(compute)], [out], fdefined, undef)

    def plevelhum(self, t, huminp, p, unit, compute, fdefined=SOME_DEFINED, undef=UNDEF, out=None):
        return self._single("mifc_plevelhum", [t, huminp], [float(p), unit.encode(), int(compute)], [out], fdefined, undef)

    def hlevelhum(self, t, huminp, ps, alevel, blevel, unit, compute, fdefined=SOME_DEFINED, undef=UNDEF, out=None):
        return self._single(
            "mifc_hlevelhum", [t, huminp, ps], [float(alevel), float(blevel), unit.encode(), int(compute)], [out], fdefined, undef
        )

    def alevelhum(self, t, huminp, p, unit, compute, fdefined=SOME_DEFINED, undef=UNDEF, out=None):
        return self._single("mifc_alevelhum", [t, huminp, p], [unit.encode(), int(compute)], [out], fdefined, undef)

    def cvhum(self, t, huminp, unit, compute, fdefined=SOME_DEFINED, undef=UNDEF, out=None):
        return self._single("mifc_cvhum", [t, huminp], [unit.encode(), int(compute)], [out], fdefined, undef)

    # -------------------------------------------------------- stencil operators
    def relvort(self, u, v, xmapr, ymapr, fdefined=SOME_DEFINED, undef=UNDEF, out=None):
        return self._single("mifc_relvort", [u, v, xmapr, ymapr], [], [out], fdefined, undef)

    def absvort(self, u, v, xmapr, ymapr, fcoriolis, fdefined=SOME_DEFINED, undef=UNDEF, out=None):
        return self._single("mifc_absvort", [u, v, xmapr, ymapr, fcoriolis], [], [out], fdefined, undef)

    def divergence(self, u, v, xmapr, ymapr, fdefined=SOME_DEFINED, undef=UNDEF, out=None):
        return self._single("mifc_divergence", [u, v, xmapr, ymapr], [], [out], fdefined, undef)

    def gradient(self, field, xmapr, ymapr, compute, fdefined=SOME_DEFINED, undef=UNDEF, out=None):
        return self._single("mifc_gradient", [field, xmapr, ymapr], [int(compute)], [out], fdefined, undef)

    def plevelgwind_xcomp(self, z, xmapr, ymapr, fcoriolis, fdefined=SOME_DEFINED, undef=UNDEF, out=None):
        return self._single("mifc_plevelgwind_xcomp", [z, xmapr, ymapr, fcoriolis], [], [out], fdefined, undef)

    def plevelgwind_ycomp(self, z, xmapr, ymapr, fcoriolis, fdefined=SOME_DEFINED, undef=UNDEF, out=None):
        return self._single("mifc_plevelgwind_ycomp", [z, xmapr, ymapr, fcoriolis], [], [out], fdefined, undef)

    def plevelgvort(self, z, xmapr, ymapr, fcoriolis, fdefined=SOME_DEFINED, undef=UNDEF, out=None):
        return self._single("mifc_plevelgvort", [z, xmapr, ymapr, fcoriolis], [], [out], fdefined, undef)

    def ilevelgwind(self, mpot, xmapr, ymapr, fcoriolis, fdefined=SOME_DEFINED, undef=UNDEF, out=(None, None)):
        return self._single("mifc_ilevelgwind", [mpot, xmapr, ymapr, fcoriolis], [], list(out), fdefined, undef, n_out=2)

    # ------------------------------------------ next operators
    def advection(self, f, u, v, xmapr, ymapr, hours, fdefined=SOME_DEFINED, undef=UNDEF, out=None):
        return self._single("mifc_advection", [f, u, v, xmapr, ymapr], [float(hours)], [out], fdefined, undef)

    def jacobian(self, field1, field2, xmapr, ymapr, fdefined=SOME_DEFINED, undef=UNDEF, out=None):
        return self._single("mifc_jacobian", [field1, field2, xmapr, ymapr], [], [out], fdefined, undef)

    def momentumXcoordinate(self, v, xmapr, fcoriolis, fcoriolisMin, fdefined=SOME_DEFINED, undef=UNDEF, out=None):
        return self._single("mifc_momentumXcoordinate", [v, xmapr, fcoriolis], [float(fcoriolisMin)], [out], fdefined, undef)

    def momentumYcoordinate(self, u, ymapr, fcoriolis, fcoriolisMin, fdefined=SOME_DEFINED, undef=UNDEF, out=None):
        return self._single("mifc_momentumYcoordinate", [u, ymapr, fcoriolis], [float(fcoriolisMin)], [out], fdefined, undef)

    def thermalFrontParameter(self, tx, xmapr, ymapr, fdefined=SOME_DEFINED, undef=UNDEF, out=None):
        return self._single("mifc_thermalFrontParameter", [tx, xmapr, ymapr], [], [out], fdefined, undef)

    def plevelqvector(self, z, t, xmapr, ymapr, fcoriolis, p, compute, fdefined=SOME_DEFINED, undef=UNDEF, out=None):
        return self._single("mifc_plevelqvector", [z, t, xmapr, ymapr, fcoriolis], [float(p), int(compute)], [out], fdefined, undef)

    # ---------------------------- the rest of the pointwise catalogue
    # Same argument order as miutil::fieldcalc; (result, flag) or None like the others.
    def plevelthe(self, t, rh, p, compute, fdefined=SOME_DEFINED, undef=UNDEF, out=None):
        return self._single("mifc_plevelthe", [t, rh], [float(p), int(compute)], [out], fdefined, undef)

    def hlevelthe(self, t, q, ps, alevel, blevel, compute, fdefined=SOME_DEFINED, undef=UNDEF, out=None):
        return self._single("mifc_hlevelthe", [t, q, ps], [float(alevel), float(blevel), int(compute)], [out], fdefined, undef)

    def alevelthe(self, t, q, p, compute, fdefined=SOME_DEFINED, undef=UNDEF, out=None):
        return self._single("mifc_alevelthe", [t, q, p], [int(compute)], [out], fdefined, undef)

    def plevelducting(self, t, h, p, compute, fdefined=SOME_DEFINED, undef=UNDEF, out=None):
        return self._single("mifc_plevelducting", [t, h], [float(p), int(compute)], [out], fdefined, undef)

    def hlevelducting(self, t, h, ps, alevel, blevel, compute, fdefined=SOME_DEFINED, undef=UNDEF, out=None):
        return self._single("mifc_hlevelducting", [t, h, ps], [float(alevel), float(blevel), int(compute)], [out], fdefined, undef)

    def alevelducting(self, t, h, p, compute, fdefined=SOME_DEFINED, undef=UNDEF, out=None):
        return self._single("mifc_alevelducting", [t, h, p], [int(compute)], [out], fdefined, undef)

    def hlevelpressure(self, ps, alevel, blevel, fdefined=SOME_DEFINED, undef=UNDEF, out=None):
        return self._single("mifc_hlevelpressure", [ps], [float(alevel), float(blevel)], [out], fdefined, undef)

    def pleveldz2tmean(self, z1, z2, p1, p2, compute, fdefined=SOME_DEFINED, undef=UNDEF, out=None):
        return self._single("mifc_pleveldz2tmean", [z1, z2], [float(p1), float(p2), int(compute)], [out], fdefined, undef)

    def kIndex(self, t500, t700, rh700, t850, rh850, p500, p700, p850, compute, fdefined=SOME_DEFINED, undef=UNDEF, out=None):
        return self._single("mifc_kIndex", [t500, t700, rh700, t850, rh850], [float(p500), float(p700), float(p850), int(compute)], [out],
                            fdefined, undef)

    def ductingIndex(self, t850, rh850, p850, compute, fdefined=SOME_DEFINED, undef=UNDEF, out=None):
        return self._single("mifc_ductingIndex", [t850, rh850], [float(p850), int(compute)], [out], fdefined, undef)

    def showalterIndex(self, t500, t850, rh850, p500, p850, compute, fdefined=SOME_DEFINED, undef=UNDEF, out=None):
        return self._single("mifc_showalterIndex", [t500, t850, rh850], [float(p500), float(p850), int(compute)], [out], fdefined, undef)

    def boydenIndex(self, t700, z700, z1000, p700, p1000, compute, fdefined=SOME_DEFINED, undef=UNDEF, out=None):
        return self._single("mifc_boydenIndex", [t700, z700, z1000], [float(p700), float(p1000), int(compute)], [out], fdefined, undef)

    def sweatIndex(self, t850, t500, td850, td500, u850, v850, u500, v500, fdefined=SOME_DEFINED, undef=UNDEF, out=None):
        return self._single("mifc_sweatIndex", [t850, t500, td850, td500, u850, v850, u500, v500], [], [out], fdefined, undef)

    def seaSoundSpeed(self, t, s, z, compute, fdefined=SOME_DEFINED, undef=UNDEF, out=None):
        return self._single("mifc_seaSoundSpeed", [t, s], [float(z), int(compute)], [out], fdefined, undef)

    def cvtemp(self, tinp, compute, fdefined=SOME_DEFINED, undef=UNDEF, out=None):
        return self._single("mifc_cvtemp", [tinp], [int(compute)], [out], fdefined, undef)

    def abshum(self, t, rhum, fdefined=SOME_DEFINED, undef=UNDEF, out=None):
        return self._single("mifc_abshum", [t, rhum], [], [out], fdefined, undef)

    def windCooling(self, t, u, v, compute, fdefined=SOME_DEFINED, undef=UNDEF, out=None):
        return self._single("mifc_windCooling", [t, u, v], [int(compute)], [out], fdefined, undef)

    def underCooledRain(self, precip, snow, tk, precipMin, snowRateMax, tcMax, fdefined=SOME_DEFINED, undef=UNDEF, out=None):
        return self._single("mifc_underCooledRain", [precip, snow, tk], [float(precipMin), float(snowRateMax), float(tcMax)], [out], fdefined, undef)

    def pressure2FlightLevel(self, pressure, fdefined=SOME_DEFINED, undef=UNDEF, out=None):
        return self._single("mifc_pressure2FlightLevel", [pressure], [], [out], fdefined, undef)

    def snow_in_cm(self, snow_water, tk2m, td2m, fdefined=SOME_DEFINED, undef=UNDEF, out=None):
        return self._single("mifc_snow_in_cm", [snow_water, tk2m, td2m], [], [out], fdefined, undef)

    def values2classes(self, fvalue, values, fdefined=SOME_DEFINED, undef=UNDEF, out=None):
        """values: the class limits (host sequence).  C order: fvalue, fclass (output), values, nvalues."""
        vals = np.ascontiguousarray(values, dtype=np.float32)
        return self._single("mifc_values2classes", [fvalue], [], [out], fdefined, undef, tail=[vals.ctypes.data, int(vals.size)])

    def shapiro2_filter(self, field, fdefined=SOME_DEFINED, undef=UNDEF, out=None):
        """Second-order Shapiro filter; pass out=field to smooth in place."""
        return self._single("mifc_shapiro2_filter", [field], [], [out], fdefined, undef)

    def vesselIcingOverland(self, airtemp, seatemp, u, v, sal, aice, fdefined=SOME_DEFINED, undef=UNDEF, out=None):
        return self._single("mifc_vesselIcingOverland", [airtemp, seatemp, u, v, sal, aice], [], [out], fdefined, undef)

    def vesselIcingMertins(self, airtemp, seatemp, u, v, sal, aice, fdefined=SOME_DEFINED, undef=UNDEF, out=None):
        return self._single("mifc_vesselIcingMertins", [airtemp, seatemp, u, v, sal, aice], [], [out], fdefined, undef)

    def winddir(self, u, v, fdefined=SOME_DEFINED, undef=UNDEF, out=None):
        """EXTENSION (no reference function): meteorological wind direction, degrees the wind blows FROM,
        dd = 270 - atan2(v, u) * 180 / pi in [0, 360), calm -> 0; see include/mifc.h."""
        return self._single("mifc_winddir", [u, v], [], [out], fdefined, undef)

    def minvalueFields(self, field1, field2, fdefined=SOME_DEFINED, undef=UNDEF, out=None):
        return self._single("mifc_minvalueFields", [field1, field2], [], [out], fdefined, undef)

    def maxvalueFields(self, field1, field2, fdefined=SOME_DEFINED, undef=UNDEF, out=None):
        return self._single("mifc_maxvalueFields", [field1, field2], [], [out], fdefined, undef)

    def minvalueFieldConst(self, field1, value, fdefined=SOME_DEFINED, undef=UNDEF, out=None):
        return self._single("mifc_minvalueFieldConst", [field1], [float(value)], [out], fdefined, undef)

    def maxvalueFieldConst(self, field1, value, fdefined=SOME_DEFINED, undef=UNDEF, out=None):
        return self._single("mifc_maxvalueFieldConst", [field1], [float(value)], [out], fdefined, undef)

    def absvalueField(self, field, fdefined=SOME_DEFINED, undef=UNDEF, out=None):
        return self._single("mifc_absvalueField", [field], [], [out], fdefined, undef)

    def log10Field(self, field, fdefined=SOME_DEFINED, undef=UNDEF, out=None):
        return self._single("mifc_log10Field", [field], [], [out], fdefined, undef)

    def pow10Field(self, field, fdefined=SOME_DEFINED, undef=UNDEF, out=None):
        return self._single("mifc_pow10Field", [field], [], [out], fdefined, undef)

    def logField(self, field, fdefined=SOME_DEFINED, undef=UNDEF, out=None):
        return self._single("mifc_logField", [field], [], [out], fdefined, undef)

    def expField(self, field, fdefined=SOME_DEFINED, undef=UNDEF, out=None):
        return self._single("mifc_expField", [field], [], [out], fdefined, undef)

    def powerField(self, field, value, fdefined=SOME_DEFINED, undef=UNDEF, out=None):
        return self._single("mifc_powerField", [field], [float(value)], [out], fdefined, undef)

    def replaceUndefined(self, field, value, fdefined=SOME_DEFINED, undef=UNDEF, out=None):
        return self._single("mifc_replaceUndefined", [field], [float(value)], [out], fdefined, undef)

    def replaceDefined(self, field, value, fdefined=SOME_DEFINED, undef=UNDEF, out=None):
        return self._single("mifc_replaceDefined", [field], [float(value)], [out], fdefined, undef)

    def fieldOPERfield(self, compute, field1, field2, fdefined=SOME_DEFINED, undef=UNDEF, out=None):
        return self._single("mifc_fieldOPERfield", [field1, field2], [], [out], fdefined, undef, lead=[int(compute)])

    def fieldOPERconstant(self, compute, field, value, fdefined=SOME_DEFINED, undef=UNDEF, out=None):
        return self._single("mifc_fieldOPERconstant", [field], [float(value)], [out], fdefined, undef, lead=[int(compute)])

    def constantOPERfield(self, compute, value, field, fdefined=SOME_DEFINED, undef=UNDEF, out=None):
        return self._single("mifc_constantOPERfield", [field], [], [out], fdefined, undef, lead=[int(compute)], pre=[float(value)])

    # ---------------------------------- reductions over ensemble members
    def _ensemble(self, name, fields, fdefined_in, lead, tail, fdefined, undef, out):
        """fields: sequence of member fields (all numpy or all CUDA tensors)."""
        fa = [_Arg(f) for f in fields]
        ref = fields[0] if len(fields) else out
        if ref is None:
            raise ValueError("no member fields and no output to take the shape from")
        ra = _Arg(ref)
        if len(ra.shape) != 2:
            return None
        nx, ny = self._nxny(ra)
        if out is None:
            out = _empty_like(ref)
        oa = _Arg(out, output=True)
        if not _same_shape(fa + [oa], ra.shape):
            return None
        mk = _memkind(fa + [oa], self.device)
        self._bind_stream(mk)
        table = (ctypes.c_void_p * max(len(fa), 1))(*[a.addr for a in fa])
        args = list(lead) + [nx, ny, ctypes.addressof(table)]
        if fdefined_in is not None:
            flags = (ctypes.c_int * max(len(fa), 1))(*[int(x) for x in fdefined_in])
            args.append(ctypes.addressof(flags))
        fd = ctypes.c_int(int(fdefined))
        args += [len(fa)] + list(tail) + [oa.addr, ctypes.addressof(fd), float(undef), mk]
        if not self._call(name, args):
            return None
        return (out if _is_torch(out) else oa.keep), fd.value

    def sumFields(self, fields, fdefined=SOME_DEFINED, undef=UNDEF, out=None):
        return self._ensemble("mifc_sumFields", fields, None, [], [], fdefined, undef, out)

    def meanValue(self, fields, fdefined_in, fdefined=SOME_DEFINED, undef=UNDEF, out=None):
        return self._ensemble("mifc_meanValue", fields, fdefined_in, [], [], fdefined, undef, out)

    def stddevValue(self, fields, fdefined_in, fdefined=SOME_DEFINED, undef=UNDEF, out=None):
        return self._ensemble("mifc_stddevValue", fields, fdefined_in, [], [], fdefined, undef, out)

    def extremeValue(self, compute, fields, fdefined=SOME_DEFINED, undef=UNDEF, out=None):
        return self._ensemble("mifc_extremeValue", fields, None, [int(compute)], [], fdefined, undef, out)

    def probability(self, compute, fields, fdefined_in, limits, fdefined=SOME_DEFINED, undef=UNDEF, out=None):
        lim = np.ascontiguousarray(limits, dtype=np.float32)
        return self._ensemble("mifc_probability", fields, fdefined_in, [int(compute)], [lim.ctypes.data, int(lim.size)], fdefined, undef, out)

    # ------------------------------------------------------------------ batched
    def vortdiv_levels(self, u, v, xmapr, ymapr, fdefined=None, undef=UNDEF, rvort=None, diverg=None, want=("rvort", "diverg")):
        """Fused relvort + divergence over u, v of shape (nlev, ny, nx).
        fdefined: int sequence per level (default SOME_DEFINED).  Returns
        ((rvort, diverg), flags ndarray) or None."""
        au, av, ax, ay = _Arg(u), _Arg(v), _Arg(xmapr), _Arg(ymapr)
        if len(au.shape) != 3:
            raise ValueError("u, v must have shape (nlev, ny, nx)")
        nlev, ny, nx = au.shape
        if rvort is None and "rvort" in want:
            rvort = _empty_like(u)
        if diverg is None and "diverg" in want:
            diverg = _empty_like(u)
        ar, ad = _Arg(rvort, allow_none=True, output=True), _Arg(diverg, allow_none=True, output=True)
        if not _same_shape([av, ar, ad], au.shape) or not _same_shape([ax, ay], (ny, nx)):
            raise ValueError("u, v, rvort, diverg must be (nlev, ny, nx) and xmapr, ymapr (ny, nx)")
        mk = _memkind([au, av, ax, ay, ar, ad], self.device)
        self._bind_stream(mk)
        flags = np.full(nlev, SOME_DEFINED, dtype=np.int32) if fdefined is None else np.array(fdefined, dtype=np.int32).reshape(nlev).copy()
        rc = self._call(
            "mifc_vortdiv_levels",
            [nx, ny, nlev, au.addr, av.addr, ax.addr, ay.addr, ar.addr, ad.addr, flags.ctypes.data, float(undef), mk],
        )
        if not rc:
            return None
        outs = tuple(o if (o is None or _is_torch(o)) else a.keep for o, a in ((rvort, ar), (diverg, ad)))
        return outs, flags

    OPS = {"relvort": 0, "absvort": 1, "divergence": 2, "vortdiv": 3, "gradient1": 4, "gradient2": 5, "gradient3": 6, "gradient4": 7,
           "plevelgwind_xcomp": 8, "plevelgwind_ycomp": 9, "plevelgvort": 10, "ilevelgwind": 11, "jacobian": 13}

    def stencil_levels(self, op, f0, f1, xmapr, ymapr, fcoriolis=None, fdefined=None, undef=UNDEF, out0=None, out1=None):
        """Any stencil operator (name from Context.OPS) over f0/f1 of shape (nlev, ny, nx).
        Returns ((out0, out1-or-None), flags) or None."""
        code = self.OPS[op]
        a0, a1 = _Arg(f0), _Arg(f1, allow_none=True)
        ax, ay, af = _Arg(xmapr, allow_none=True), _Arg(ymapr, allow_none=True), _Arg(fcoriolis, allow_none=True)
        nlev, ny, nx = a0.shape
        two = code in (3, 11)
        if out0 is None:
            out0 = _empty_like(f0)
        if two and out1 is None:
            out1 = _empty_like(f0)
        o0, o1 = _Arg(out0, output=True), _Arg(out1 if two else None, allow_none=True, output=True)
        if len(a0.shape) != 3 or not _same_shape([a1, o0, o1], a0.shape) or not _same_shape([ax, ay, af], (ny, nx)):
            raise ValueError("level fields must be (nlev, ny, nx) and the map / Coriolis fields (ny, nx)")
        mk = _memkind([a0, a1, ax, ay, af, o0, o1], self.device)
        self._bind_stream(mk)
        flags = np.full(nlev, SOME_DEFINED, dtype=np.int32) if fdefined is None else np.array(fdefined, dtype=np.int32).reshape(nlev).copy()
        rc = self._call("mifc_stencil_levels", [code, nx, ny, nlev, a0.addr, a1.addr, ax.addr, ay.addr, af.addr, o0.addr, o1.addr,
                                                flags.ctypes.data, float(undef), mk])
        if not rc:
            return None
        r0 = out0 if _is_torch(out0) else o0.keep
        r1 = None if not two else (out1 if _is_torch(out1) else o1.keep)
        return (r0, r1), flags

    OPS_EX = {"advection": 12, "thermalFrontParameter": 14, "plevelqvector": 15, "shapiro2_filter": 17}

    def stencil_levels_ex(self, op, f0, f1=None, f2=None, xmapr=None, ymapr=None, fcoriolis=None, level_scalars=None, scalar=0.0, compute=0,
                          fdefined=None, undef=UNDEF, out0=None):
        """The f1 operators over a level batch (mifc_stencil_levels_ex): op in OPS_EX; fields (nlev, ny, nx),
        map / Coriolis fields (ny, nx) shared.  advection: f0 = f, f1 = u, f2 = v, scalar = hours;
        thermalFrontParameter: f0 = tx; plevelqvector: f0 = z, f1 = t, level_scalars = p per level, compute 1..4;
        shapiro2_filter: f0 = field.  Returns (out0, flags) or None."""
        code = self.OPS_EX[op]
        a0, a1, a2 = _Arg(f0), _Arg(f1, allow_none=True), _Arg(f2, allow_none=True)
        ax, ay, af = _Arg(xmapr, allow_none=True), _Arg(ymapr, allow_none=True), _Arg(fcoriolis, allow_none=True)
        if len(a0.shape) != 3:
            raise ValueError("level fields must be (nlev, ny, nx)")
        nlev, ny, nx = a0.shape
        if out0 is None:
            out0 = _empty_like(f0)
        o0 = _Arg(out0, output=True)
        if not _same_shape([a1, a2, o0], a0.shape) or not _same_shape([ax, ay, af], (ny, nx)):
            raise ValueError("level fields must be (nlev, ny, nx) and the map / Coriolis fields (ny, nx)")
        mk = _memkind([a0, a1, a2, ax, ay, af, o0], self.device)
        self._bind_stream(mk)
        flags = np.full(nlev, SOME_DEFINED, dtype=np.int32) if fdefined is None else np.array(fdefined, dtype=np.int32).reshape(nlev).copy()
        ls = None if level_scalars is None else np.ascontiguousarray(level_scalars, dtype=np.float32).reshape(nlev)
        rc = self._call("mifc_stencil_levels_ex", [code, nx, ny, nlev, a0.addr, a1.addr, a2.addr, ax.addr, ay.addr, af.addr,
                                                   None if ls is None else ls.ctypes.data, float(scalar), int(compute), o0.addr, None,
                                                   flags.ctypes.data, float(undef), mk])
        if not rc:
            return None
        return (out0 if _is_torch(out0) else o0.keep), flags

    def vortdiv_levels_enqueue(self, u, v, xmapr, ymapr, rvort, diverg, fdefined=None, undef=UNDEF, n_undefined=None):
        """Asynchronous form on device tensors; n_undefined: int64 CUDA tensor[nlev] or None.
        u, v and rvort, diverg may be level-padded batches (see batch_empty)."""
        au, av, ax, ay = _Arg(u, levels=True), _Arg(v, levels=True), _Arg(xmapr), _Arg(ymapr)
        ar, ad = _Arg(rvort, allow_none=True, levels=True), _Arg(diverg, allow_none=True, levels=True)
        if len(au.shape) != 3:
            raise ValueError("u, v must have shape (nlev, ny, nx)")
        nlev, ny, nx = au.shape
        if not _same_shape([av, ar, ad], au.shape) or not _same_shape([ax, ay], (ny, nx)):
            raise ValueError("u, v, rvort, diverg must be (nlev, ny, nx) and xmapr, ymapr (ny, nx)")
        if _memkind([au, av, ax, ay, ar, ad], self.device) != MEM_DEVICE:
            raise ValueError("the *_enqueue calls take device tensors only")
        outs = [a for a in (ar, ad) if a.addr is not None]
        if not outs:
            raise ValueError("at least one of rvort, diverg is required")
        if av.lstride != au.lstride or (len(outs) == 2 and outs[0].lstride != outs[1].lstride):
            raise ValueError("u and v (and rvort and diverg) must share one level stride")
        flags = None if fdefined is None else np.array(fdefined, dtype=np.int32).reshape(nlev).copy()
        self._bind_stream(MEM_DEVICE)
        rc = self._call(
            "mifc_vortdiv_levels_strided_enqueue",
            [
                nx, ny, nlev, au.addr, av.addr, ax.addr, ay.addr, ar.addr, ad.addr, au.lstride, outs[0].lstride,
                flags, float(undef),
                None if n_undefined is None else n_undefined.data_ptr(),
            ],
        )
        return bool(rc)

    def vortdiv_ff_levels_enqueue(self, u, v, xmapr, ymapr, rvort, diverg, ff, fdefined=None, undef=UNDEF, n_undefined=None, n_undefined_ff=None):
        """Vorticity, divergence and the wind speed of a level batch in one pass (mifc_vortdiv_ff_levels_enqueue); device tensors
        (nlev, ny, nx); n_undefined / n_undefined_ff: int64[nlev] (classify against nx*ny - 2*nx / nx*ny), None allowed when every
        level is ALL_DEFINED."""
        a = [_Arg(x) for x in (u, v, xmapr, ymapr, rvort, diverg, ff)]
        if len(a[0].shape) != 3:
            raise ValueError("u, v must have shape (nlev, ny, nx)")
        nlev, ny, nx = a[0].shape
        if not _same_shape([a[1], a[4], a[5], a[6]], a[0].shape) or not _same_shape(a[2:4], (ny, nx)):
            raise ValueError("u, v, rvort, diverg, ff must be (nlev, ny, nx) and xmapr, ymapr (ny, nx)")
        if _memkind(a, self.device) != MEM_DEVICE:
            raise ValueError("the *_enqueue calls take device tensors only")
        flags = None if fdefined is None else np.array(fdefined, dtype=np.int32).reshape(nlev).copy()
        self._bind_stream(MEM_DEVICE)
        rc = self._call("mifc_vortdiv_ff_levels_enqueue", [nx, ny, nlev] + [x.addr for x in a] + [
            flags, float(undef), None if n_undefined is None else n_undefined.data_ptr(),
            None if n_undefined_ff is None else n_undefined_ff.data_ptr()])
        return bool(rc)

    def stencil_levels_enqueue(self, op, f0, f1, xmapr, ymapr, fcoriolis, out0, out1=None, fdefined=None, undef=UNDEF, n_undefined=None):
        """Asynchronous form of stencil_levels on device tensors (mifc_stencil_levels_enqueue): nothing is read back;
        n_undefined: int64 CUDA tensor[nlev] (None allowed when every level is ALL_DEFINED).  The flag of level l is
        classify(n_undefined[l], stencil_count_domain(op, nx, ny)) once the stream has drained."""
        code = self.OPS[op]
        a0, a1 = _Arg(f0), _Arg(f1, allow_none=True)
        ax, ay, af = _Arg(xmapr), _Arg(ymapr), _Arg(fcoriolis, allow_none=True)
        o0, o1 = _Arg(out0, allow_none=True, output=True), _Arg(out1, allow_none=True, output=True)
        if len(a0.shape) != 3:
            raise ValueError("level fields must be (nlev, ny, nx)")
        nlev, ny, nx = a0.shape
        if not _same_shape([a1, o0, o1], a0.shape) or not _same_shape([ax, ay, af], (ny, nx)):
            raise ValueError("level fields must be (nlev, ny, nx) and the map / Coriolis fields (ny, nx)")
        if _memkind([a0, a1, ax, ay, af, o0, o1], self.device) != MEM_DEVICE:
            raise ValueError("the *_enqueue calls take device tensors only")
        flags = None if fdefined is None else np.array(fdefined, dtype=np.int32).reshape(nlev).copy()
        self._bind_stream(MEM_DEVICE)
        rc = self._call("mifc_stencil_levels_enqueue", [code, nx, ny, nlev, a0.addr, a1.addr, ax.addr, ay.addr, af.addr, o0.addr, o1.addr,
                                                        flags, float(undef),
                                                        None if n_undefined is None else n_undefined.data_ptr()])
        return bool(rc)

    def last_stencil_form(self):
        """The kernel form this thread's last stencil launch took (mifc_last_stencil_form; a diagnostic for tests)."""
        return self._lib.mifc_last_stencil_form().decode()

    def stencil_count_domain(self, op, nx, ny):
        return int(self._lib.mifc_stencil_count_domain(self.OPS[op], nx, ny))

    def batch_level_stride(self, nx, ny):
        """Recommended distance (floats) between the levels of a device-resident batch (mifc_batch_level_stride)."""
        return int(self._lib.mifc_batch_level_stride(int(nx), int(ny)))

    def batch_empty(self, nlev, ny, nx, level_stride=None):
        """Uninitialised (nlev, ny, nx) float32 device batch whose levels are level_stride floats apart
        (default: batch_level_stride).  The padding between levels is never read or written."""
        import torch

        ls = self.batch_level_stride(nx, ny) if level_stride is None else int(level_stride)
        if ls < ny * nx or ls % 4 != 0:
            raise ValueError("level stride must be a multiple of 4 and at least ny*nx")
        base = torch.empty(nlev * ls, dtype=torch.float32, device=torch.device("cuda", self.device))
        return torch.as_strided(base, (nlev, ny, nx), (ls, nx, 1))

    def hlevel_derived_levels(self, u, v, t, q, ps, alevel, blevel, fdef_wind=None, fdef_thermo=None, undef=UNDEF,
                              want=("ff", "rh", "theta"), out=None):
        """Fused ff / RH(%) / theta on hybrid levels; u, v, t, q: (nlev, ny, nx), ps: (ny, nx).
        Returns ({name: array}, {name: flags}) or None."""
        ref = next(x for x in (u, t) if x is not None)
        nlev, ny, nx = _Arg(ref).shape
        out = dict(out or {})
        for k in want:
            if out.get(k) is None:
                out[k] = _empty_like(ref)
        a = {k: _Arg(x, allow_none=True) for k, x in dict(u=u, v=v, t=t, q=q, ps=ps).items()}
        o = {k: _Arg(out.get(k), allow_none=True) for k in ("ff", "rh", "theta")}
        if not _same_shape([a[k] for k in ("u", "v", "t", "q")] + list(o.values()), (nlev, ny, nx)) or not _same_shape([a["ps"]], (ny, nx)):
            raise ValueError("level fields must be (nlev, ny, nx) and ps (ny, nx)")
        mk = _memkind(list(a.values()) + list(o.values()), self.device)
        self._bind_stream(mk)
        al = np.ascontiguousarray(alevel if alevel is not None else np.zeros(nlev), dtype=np.float32).reshape(nlev)
        bl = np.ascontiguousarray(blevel if blevel is not None else np.ones(nlev), dtype=np.float32).reshape(nlev)
        fw = np.full(nlev, SOME_DEFINED, np.int32) if fdef_wind is None else np.array(fdef_wind, np.int32).reshape(nlev).copy()
        ft = np.full(nlev, SOME_DEFINED, np.int32) if fdef_thermo is None else np.array(fdef_thermo, np.int32).reshape(nlev).copy()
        fo = {k: np.full(nlev, -1, np.int32) for k in ("ff", "rh", "theta")}
        rc = self._call(
            "mifc_hlevel_derived_levels",
            [
                nx, ny, nlev, a["u"].addr, a["v"].addr, a["t"].addr, a["q"].addr, a["ps"].addr, al, bl,
                o["ff"].addr, o["rh"].addr, o["theta"].addr, fw, ft,
                fo["ff"].ctypes.data, fo["rh"].ctypes.data, fo["theta"].ctypes.data, float(undef), mk,
            ],
        )
        if not rc:
            return None
        res = {k: (out[k] if _is_torch(out[k]) else o[k].keep) for k in want}
        return res, {k: fo[k] for k in want}

    def hlevel_derived_batch(self, u, v, t, h, ps, alevel, blevel, temp=None, hum=None, hum2=None, ff=True, dd=False, fdef_wind=None,
                             fdef_thermo=None, undef=UNDEF, out=None, enqueue_counts=None):
        """The general fused derived batch (mifc_hlevel_derived_batch): per level any of
          ff   = vectorabs(u, v)                         (ff=True)
          temp = hleveltemp(t, ps, a, b, unit, compute)  (temp=(unit, compute))
          hum  = hlevelhum(t, h, ps, a, b, unit, compute)   (hum=(unit, compute))
          hum2 = a second hlevelhum variant of the same inputs (hum2=(unit, compute))
          dd   = wind direction from u, v (dd=True) -- EXTENSION, not a reference function
        u, v, t, h: (nlev, ny, nx); ps: (ny, nx).  out: optional dict of preallocated outputs.
        Returns ({name: array}, {name: flags}) or None.  With enqueue_counts (int64 CUDA tensor[5*nlev]) the call is
        asynchronous on device tensors and returns the outputs only (counts: ff | temp | hum | hum2 | dd)."""
        ref = next(x for x in (u, t) if x is not None)
        nlev, ny, nx = _Arg(ref).shape
        names = [k for k, w in (("ff", ff), ("temp", temp), ("hum", hum), ("hum2", hum2), ("dd", dd)) if w]
        out = dict(out or {})
        for k in names:
            if out.get(k) is None:
                out[k] = _empty_like(ref)
        a = {k: _Arg(x, allow_none=True) for k, x in dict(u=u, v=v, t=t, h=h, ps=ps).items()}
        o = {k: _Arg(out.get(k) if k in names else None, allow_none=True, output=True) for k in ("ff", "temp", "hum", "hum2", "dd")}
        if not _same_shape([a[k] for k in ("u", "v", "t", "h")] + list(o.values()), (nlev, ny, nx)) or not _same_shape([a["ps"]], (ny, nx)):
            raise ValueError("level fields must be (nlev, ny, nx) and ps (ny, nx)")
        mk = _memkind(list(a.values()) + list(o.values()), self.device)
        self._bind_stream(mk)
        al = np.ascontiguousarray(alevel if alevel is not None else np.zeros(nlev), dtype=np.float32).reshape(nlev)
        bl = np.ascontiguousarray(blevel if blevel is not None else np.ones(nlev), dtype=np.float32).reshape(nlev)
        fw = np.full(nlev, SOME_DEFINED, np.int32) if fdef_wind is None else np.array(fdef_wind, np.int32).reshape(nlev).copy()
        ft = np.full(nlev, SOME_DEFINED, np.int32) if fdef_thermo is None else np.array(fdef_thermo, np.int32).reshape(nlev).copy()
        unit = lambda w: (w[0] if w else "").encode()
        comp = lambda w: int(w[1]) if w else 0
        common = [nx, ny, nlev, a["u"].addr, a["v"].addr, a["t"].addr, a["h"].addr, a["ps"].addr, al, bl,
                  o["ff"].addr, o["temp"].addr, unit(temp), comp(temp), o["hum"].addr, unit(hum), comp(hum), o["hum2"].addr, unit(hum2), comp(hum2),
                  o["dd"].addr, fw, ft]
        if enqueue_counts is not None:
            if mk != MEM_DEVICE:
                raise ValueError("the *_enqueue calls take device tensors only")
            rc = self._call("mifc_hlevel_derived_batch_enqueue", common + [float(undef), enqueue_counts.data_ptr()])
            return {k: out[k] for k in names} if rc else None
        fo = {k: np.full(nlev, -1, np.int32) for k in ("ff", "temp", "hum", "hum2", "dd")}
        rc = self._call("mifc_hlevel_derived_batch", common + [fo[k].ctypes.data for k in ("ff", "temp", "hum", "hum2", "dd")] + [float(undef), mk])
        if not rc:
            return None
        res = {k: (out[k] if _is_torch(out[k]) else o[k].keep) for k in names}
        return res, {k: fo[k] for k in names}

    def hlevel_derived_levels_enqueue(self, u, v, t, q, ps, alevel, blevel, ff, rh, theta, n_undefined, fdef_wind=None,
                                      fdef_thermo=None, undef=UNDEF):
        ref = next(x for x in (u, t) if x is not None)
        nlev, ny, nx = _Arg(ref).shape
        a = [_Arg(x, allow_none=True) for x in (u, v, t, q, ps)]
        o = [_Arg(x, allow_none=True) for x in (ff, rh, theta)]
        al = np.ascontiguousarray(alevel, dtype=np.float32).reshape(nlev)
        bl = np.ascontiguousarray(blevel, dtype=np.float32).reshape(nlev)
        fw = None if fdef_wind is None else np.array(fdef_wind, np.int32).reshape(nlev).copy()
        ft = None if fdef_thermo is None else np.array(fdef_thermo, np.int32).reshape(nlev).copy()
        if not _same_shape(a[:4] + o, (nlev, ny, nx)) or not _same_shape([a[4]], (ny, nx)):
            raise ValueError("level fields must be (nlev, ny, nx) and ps (ny, nx)")
        if _memkind(a + o, self.device) != MEM_DEVICE:
            raise ValueError("the *_enqueue calls take device tensors only")
        self._bind_stream(MEM_DEVICE)
        rc = self._call(
            "mifc_hlevel_derived_levels_enqueue",
            [nx, ny, nlev] + [x.addr for x in a] + [al, bl] + [x.addr for x in o]
            + [fw, ft, float(undef), n_undefined.data_ptr()],
        )
        return bool(rc)

    def vortdiv_slab_enqueue(self, nx, ny_global, j0, ny_local, u_halo, v_halo, xmapr, ymapr, rvort, diverg, fdefined_in=SOME_DEFINED,
                             undef=UNDEF, n_undefined=None, rows=None, accumulate=False):
        """Row-slab form (see include/mifc.h); all tensors on the device.  rows=(begin, end) restricts the
        launch to those owned rows (halo overlap); accumulate=True adds to n_undefined instead of zeroing it."""
        a = [_Arg(x, allow_none=True) for x in (u_halo, v_halo, xmapr, ymapr, rvort, diverg)]
        if not _same_shape(a[:2], (ny_local + 2, nx)) or not _same_shape(a[2:], (ny_local, nx)):
            raise ValueError("u_halo, v_halo must be (ny_local + 2, nx); xmapr, ymapr, rvort, diverg (ny_local, nx)")
        if _memkind(a, self.device) != MEM_DEVICE:
            raise ValueError("the *_enqueue calls take device tensors only")
        self._bind_stream(MEM_DEVICE)
        r0, r1 = (0, int(ny_local)) if rows is None else (int(rows[0]), int(rows[1]))
        rc = self._call(
            "mifc_vortdiv_slab_rows_enqueue",
            [int(nx), int(ny_global), int(j0), int(ny_local), r0, r1] + [x.addr for x in a]
            + [int(fdefined_in), float(undef), None if n_undefined is None else n_undefined.data_ptr(), 1 if accumulate else 0],
        )
        return bool(rc)

    # ---- a sequence of *_enqueue calls as one launch (include/mifc.h: mifc_graph_*)
    def counts_accumulate(self, on=True):
        """The *_enqueue entries add to the counters they are given instead of zeroing them first (mifc_counts_accumulate)."""
        return bool(self._lib.mifc_counts_accumulate(self._ctx, 1 if on else 0))

    def zero_counts_enqueue(self, counts):
        """One asynchronous fill of an int64 CUDA tensor of counters (mifc_zero_counts_enqueue)."""
        self._bind_stream(MEM_DEVICE)
        return bool(self._call("mifc_zero_counts_enqueue", [counts.data_ptr(), counts.numel()]))

    def graph_capture(self, max_levels_per_call=0, lanes=1):
        """with ctx.graph_capture() as g: ... *_enqueue calls ...   ->  g.launch() replays them with one runtime call.
        lanes > 1: g.lane(k) before a call records it in lane k; calls in different lanes must be independent."""
        return Graph(self, max_levels_per_call, lanes)

    # ---- the decomposed step as one call (include/mifc.h: mifc_comm_*, mifc_slab_plan_*)
    def comm_unique_id(self):
        """mifc_comm_unique_id: the bytes rank 0 hands to every rank of a new communicator."""
        ident = ctypes.create_string_buffer(128)
        if not self._lib.mifc_comm_unique_id(ctypes.cast(ident, ctypes.c_void_p)):
            raise RuntimeError("mifc_comm_unique_id failed (RCCL not loadable?)")
        return ident.raw

    def comm_init(self, ident, rank, world):
        """mifc_comm_init: a communicator of the library's own for `world` processes, this one being `rank`."""
        buf = ctypes.create_string_buffer(bytes(ident), 128)
        if not self._call("mifc_comm_init", [ctypes.cast(buf, ctypes.c_void_p), int(rank), int(world)]):
            raise RuntimeError("mifc_comm_init: " + self.last_error())
        return True

    def comm_init_from_torch(self, group=None):
        """A communicator for the ranks of a torch.distributed group: rank 0's id is broadcast through the group."""
        import torch.distributed as dist

        rank, world = dist.get_rank(group), dist.get_world_size(group)
        box = [self.comm_unique_id() if rank == 0 else None]
        dist.broadcast_object_list(box, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
        return self.comm_init(box[0], rank, world)

    def comm_adopt(self, nccl_comm_ptr):
        """Use an ncclComm_t the caller owns (an integer address, e.g. ProcessGroupNCCL._comm_ptr())."""
        return bool(self._call("mifc_comm_adopt", [int(nccl_comm_ptr)]))

    def comm_release(self):
        return bool(self._lib.mifc_comm_release(self._ctx))

    def comm_info(self):
        """(has a communicator, rank, world)"""
        r, w = ctypes.c_int(0), ctypes.c_int(1)
        has = self._lib.mifc_comm_info(self._ctx, ctypes.cast(ctypes.pointer(r), ctypes.c_void_p), ctypes.cast(ctypes.pointer(w), ctypes.c_void_p))
        return bool(has), r.value, w.value

    def slab_plan(self, nx, ny_global, j0, ny_local, u_halo, v_halo, xmapr, ymapr, rvort, diverg, fdefined_in=SOME_DEFINED, undef=UNDEF,
                  n_undefined=None):
        """mifc_slab_plan_create on device tensors: u_halo, v_halo (nlev, ny_local + 2, nx) or (ny_local + 2, nx); xmapr, ymapr
        (ny_local, nx); rvort, diverg (nlev, ny_local, nx) or (ny_local, nx); n_undefined int64[nlev].  -> SlabPlan"""
        return SlabPlan(self, nx, ny_global, j0, ny_local, u_halo, v_halo, xmapr, ymapr, rvort, diverg, fdefined_in, undef, n_undefined)

    def halo_copy_enqueue(self, dst, src_ctx, src):
        """dst (tensor on this context's device) <- src (tensor on src_ctx's device), see mifc_halo_copy_enqueue."""
        if dst.numel() != src.numel() or not dst.is_contiguous() or not src.is_contiguous():
            raise ValueError("halo rows must be contiguous and of equal length")
        return bool(self._call("mifc_halo_copy_enqueue", [dst.data_ptr(), src_ctx._ctx, src.data_ptr(), dst.numel()]))

    def diag_division(self, a, b, g, shared, plain):
        """Measurement build only (include/mifc_measure.h): arithmetic self-check; device tensors of equal length."""
        self._measure_entry("mifc_diag_division")
        self._bind_stream(MEM_DEVICE)
        return bool(self._call("mifc_diag_division", [a.data_ptr(), b.data_ptr(), g.data_ptr(), shared.data_ptr(), plain.data_ptr(), a.numel()]))

    def bench_stream2(self, variant, blocks, dst0, dst1, src0, src1):
        """Measurement build only (include/mifc_measure.h): bandwidth yardstick; device tensors."""
        self._measure_entry("mifc_bench_stream2")
        self._bind_stream(MEM_DEVICE)
        n = src0.numel()
        return bool(self._call("mifc_bench_stream2", [int(variant), int(blocks), dst0.data_ptr(), dst1.data_ptr(), src0.data_ptr(), src1.data_ptr(), n]))


class SlabPlan:
    """One rank's row slab of a horizontally decomposed level batch, bound to its buffers (mifc_slab_plan_*): step() enqueues
    the whole decomposed step -- RCCL halo exchange, interior rows meanwhile, boundary strips, count all-reduce -- as one
    call that replays a HIP graph; begin() / finish() bracket a halo exchange the caller does itself."""

    def __init__(self, ctx, nx, ny_global, j0, ny_local, u_halo, v_halo, xmapr, ymapr, rvort, diverg, fdefined_in, undef, n_undefined):
        a = [_Arg(x, allow_none=True) for x in (u_halo, v_halo, xmapr, ymapr, rvort, diverg)]
        nlev = u_halo.shape[0] if u_halo.dim() == 3 else 1
        halo_shape = (nlev, ny_local + 2, nx) if u_halo.dim() == 3 else (ny_local + 2, nx)
        out_shape = (nlev, ny_local, nx) if u_halo.dim() == 3 else (ny_local, nx)
        if not _same_shape(a[:2], halo_shape) or not _same_shape(a[2:4], (ny_local, nx)) or not _same_shape(a[4:], out_shape):
            raise ValueError("u_halo, v_halo must be ([nlev,] ny_local + 2, nx); xmapr, ymapr (ny_local, nx); rvort, diverg ([nlev,] ny_local, nx)")
        if _memkind(a, ctx.device) != MEM_DEVICE:
            raise ValueError("a slab plan takes device tensors only")
        if n_undefined is not None and n_undefined.numel() < nlev:
            raise ValueError("n_undefined needs one int64 per level")
        self._ctx, self._keep = ctx, (u_halo, v_halo, xmapr, ymapr, rvort, diverg, n_undefined)
        ctx._bind_stream(MEM_DEVICE)
        self._plan = ctx._lib.mifc_slab_plan_create(ctx._ctx, int(nx), int(ny_global), int(j0), int(ny_local), int(nlev), *[x.addr for x in a],
                                                    int(fdefined_in), float(undef), None if n_undefined is None else n_undefined.data_ptr())
        if not self._plan:
            raise RuntimeError("mifc_slab_plan_create: " + (ctx.last_error() or "invalid arguments"))

    def _run(self, name):
        self._ctx._bind_stream(MEM_DEVICE)
        if not getattr(self._ctx._lib, name)(self._plan):
            raise RuntimeError("%s: %s" % (name, self._ctx.last_error()))
        return True

    def step(self):
        return self._run("mifc_slab_plan_step")

    def begin(self):
        return self._run("mifc_slab_plan_begin")

    def finish(self):
        return self._run("mifc_slab_plan_finish")

    @property
    def uses_graph(self):
        return bool(self._ctx._lib.mifc_slab_plan_uses_graph(self._plan))

    def close(self):
        if self._plan:
            self._ctx._lib.mifc_slab_plan_destroy(self._plan)
            self._plan = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Graph:
    """A recorded sequence of asynchronous calls (mifc_graph_begin / _end / _launch).  The tensors passed to the recorded
    calls must stay alive and in place for as long as the graph is launched: the graph holds their addresses."""

    def __init__(self, ctx, max_levels_per_call=0, lanes=1):
        self._ctx, self._graph, self._max, self._lanes = ctx, None, int(max_levels_per_call), int(lanes)

    def __enter__(self):
        self._ctx._bind_stream(MEM_DEVICE)
        self._ctx._frozen_stream = True  # the recorded calls must not re-bind the stream
        if not self._ctx._lib.mifc_graph_begin_lanes(self._ctx._ctx, self._max, self._lanes):
            self._ctx._frozen_stream = False
            raise RuntimeError("mifc_graph_begin: " + self._ctx.last_error())
        return self

    def __exit__(self, *exc):
        self._graph = self._ctx._lib.mifc_graph_end(self._ctx._ctx)
        self._ctx._frozen_stream = False
        if exc[0] is None and not self._graph:
            raise RuntimeError("mifc_graph_end: " + self._ctx.last_error())
        return False

    def lane(self, k):
        if not self._ctx._lib.mifc_graph_lane(self._ctx._ctx, int(k)):
            raise RuntimeError("mifc_graph_lane: " + self._ctx.last_error())

    def launch(self):
        self._ctx._bind_stream(MEM_DEVICE)
        if not self._ctx._lib.mifc_graph_launch(self._graph):
            raise RuntimeError("mifc_graph_launch: " + self._ctx.last_error())
        return True

    def close(self):
        if self._graph:
            self._ctx._lib.mifc_graph_destroy(self._graph)
            self._graph = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class PreparedCalls:
    """The C calls of a recorded wrapper sequence (Context.prepare), ready to be repeated: launch() is one ctypes call per
    recorded call, with the arguments converted once."""

    def __init__(self, ctx, recorded):
        self._ctx = ctx
        self._keep = [k for _, _, _, k in recorded]
        self._calls = []
        for name, fn, cargs, _ in recorded:
            conv = tuple(t(a) if a is not None else None for t, a in zip(fn.argtypes[1:], cargs))
            self._calls.append((name, fn, conv))
        self._ctxp = ctypes.c_void_p(ctx._ctx) if not isinstance(ctx._ctx, ctypes.c_void_p) else ctx._ctx

    def launch(self):
        c = self._ctxp
        for name, fn, cargs in self._calls:
            if not fn(c, *cargs):
                raise RuntimeError("%s: %s" % (name, self._ctx.last_error()))
        return True

    def __len__(self):
        return len(self._calls)
