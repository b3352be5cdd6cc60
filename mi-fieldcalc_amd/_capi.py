"""ctypes binding of the C ABI declared in include/mifc.h.

Loading never falls back to anything: if ``libmifc.so`` (the HIP build for
gfx950) is missing, importing this module raises.  Computing additionally needs
a GPU: ``mifc_create`` returns NULL without one and ``Context`` raises.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("MIFC_LIB_PATH") or os.path.join(_HERE, "libmifc.so")  # override: experiments in tools/ only

c_float_p = ctypes.POINTER(ctypes.c_float)
c_int_p = ctypes.POINTER(ctypes.c_int)
c_u64_p = ctypes.POINTER(ctypes.c_ulonglong)

_T = {
    "ctx": ctypes.c_void_p,
    "i": ctypes.c_int,
    "f": ctypes.c_float,
    "fl": ctypes.c_float,  # float return value
    "s": ctypes.c_char_p,
    "p": ctypes.c_void_p,  # float* (host or device), passed as an address
    "pi": ctypes.c_void_p,  # int*  (host)
    "pu": ctypes.c_void_p,  # unsigned long long* (device)
    "z": ctypes.c_size_t,
    "u64": ctypes.c_ulonglong,
}

# name -> (restype, [argument kinds]); mirrors include/mifc.h one to one
SIGNATURES = {
    "mifc_abi_version": ("i", []),
    "mifc_device_count": ("i", []),
    "mifc_create": ("ctx", ["i"]),
    "mifc_destroy": (None, ["ctx"]),
    "mifc_last_error": ("s", ["ctx"]),
    "mifc_set_stream": ("i", ["ctx", "p"]),
    "mifc_use_own_stream": ("i", ["ctx"]),
    "mifc_synchronize": ("i", ["ctx"]),
    "mifc_reload_env": ("i", ["ctx"]),
    "mifc_not_built": ("i", ["ctx", "s"]),
    "mifc_device_alloc": ("p", ["ctx", "z"]),
    "mifc_device_free": ("i", ["ctx", "p"]),
    "mifc_copy_to_device": ("i", ["ctx", "p", "p", "z"]),
    "mifc_copy_to_host": ("i", ["ctx", "p", "p", "z"]),
    "mifc_batch_alloc_placed": ("i", ["ctx", "i", "z", "i", "z", "i", "i", "i", "p", "p", "p", "p"]),
    "mifc_batch_free_placed": ("i", ["ctx", "p", "i"]),
    "mifc_hold_field": ("i", ["ctx", "p", "z"]),
    "mifc_release_field": ("i", ["ctx", "p"]),
    "mifc_classify": ("i", ["u64", "u64"]),
    "mifc_counts_accumulate": ("i", ["ctx", "i"]),
    "mifc_zero_counts_enqueue": ("i", ["ctx", "pu", "z"]),
    # elementwise
    "mifc_vectorabs": ("i", ["ctx", "i", "i", "p", "p", "p", "pi", "f", "i"]),
    "mifc_pleveltemp": ("i", ["ctx", "i", "i", "p", "f", "s", "i", "p", "pi", "f", "i"]),
    "mifc_hleveltemp": ("i", ["ctx", "i", "i", "p", "p", "f", "f", "s", "i", "p", "pi", "f", "i"]),
    "mifc_aleveltemp": ("i", ["ctx", "i", "i", "p", "p", "s", "i", "p", "pi", "f", "i"]),
    "mifc_plevelhum": ("i", ["ctx", "i", "i", "p", "p", "f", "s", "i", "p", "pi", "f", "i"]),
    "mifc_hlevelhum": ("i", ["ctx", "i", "i", "p", "p", "p", "f", "f", "s", "i", "p", "pi", "f", "i"]),
    "mifc_alevelhum": ("i", ["ctx", "i", "i", "p", "p", "p", "s", "i", "p", "pi", "f", "i"]),
    "mifc_cvhum": ("i", ["ctx", "i", "i", "p", "p", "s", "i", "p", "pi", "f", "i"]),
    # stencils
    "mifc_relvort": ("i", ["ctx", "i", "i", "p", "p", "p", "p", "p", "pi", "f", "i"]),
    "mifc_absvort": ("i", ["ctx", "i", "i", "p", "p", "p", "p", "p", "p", "pi", "f", "i"]),
    "mifc_divergence": ("i", ["ctx", "i", "i", "p", "p", "p", "p", "p", "pi", "f", "i"]),
    "mifc_gradient": ("i", ["ctx", "i", "i", "p", "p", "p", "i", "p", "pi", "f", "i"]),
    "mifc_plevelgwind_xcomp": ("i", ["ctx", "i", "i", "p", "p", "p", "p", "p", "pi", "f", "i"]),
    "mifc_plevelgwind_ycomp": ("i", ["ctx", "i", "i", "p", "p", "p", "p", "p", "pi", "f", "i"]),
    "mifc_plevelgvort": ("i", ["ctx", "i", "i", "p", "p", "p", "p", "p", "pi", "f", "i"]),
    "mifc_ilevelgwind": ("i", ["ctx", "i", "i", "p", "p", "p", "p", "p", "p", "pi", "f", "i"]),
    # SURVEY.md 8f-1
    "mifc_advection": ("i", ["ctx", "i", "i", "p", "p", "p", "p", "p", "f", "p", "pi", "f", "i"]),
    "mifc_jacobian": ("i", ["ctx", "i", "i", "p", "p", "p", "p", "p", "pi", "f", "i"]),
    "mifc_momentumXcoordinate": ("i", ["ctx", "i", "i", "p", "p", "p", "f", "p", "pi", "f", "i"]),
    "mifc_momentumYcoordinate": ("i", ["ctx", "i", "i", "p", "p", "p", "f", "p", "pi", "f", "i"]),
    "mifc_thermalFrontParameter": ("i", ["ctx", "i", "i", "p", "p", "p", "p", "pi", "f", "i"]),
    "mifc_plevelqvector": ("i", ["ctx", "i", "i", "p", "p", "p", "p", "p", "f", "i", "p", "pi", "f", "i"]),
    # the rest of the pointwise catalogue
    "mifc_plevelthe": ("i", ["ctx", "i", "i", "p", "p", "f", "i", "p", "pi", "f", "i"]),
    "mifc_hlevelthe": ("i", ["ctx", "i", "i", "p", "p", "p", "f", "f", "i", "p", "pi", "f", "i"]),
    "mifc_alevelthe": ("i", ["ctx", "i", "i", "p", "p", "p", "i", "p", "pi", "f", "i"]),
    "mifc_plevelducting": ("i", ["ctx", "i", "i", "p", "p", "f", "i", "p", "pi", "f", "i"]),
    "mifc_hlevelducting": ("i", ["ctx", "i", "i", "p", "p", "p", "f", "f", "i", "p", "pi", "f", "i"]),
    "mifc_alevelducting": ("i", ["ctx", "i", "i", "p", "p", "p", "i", "p", "pi", "f", "i"]),
    "mifc_hlevelpressure": ("i", ["ctx", "i", "i", "p", "f", "f", "p", "pi", "f", "i"]),
    "mifc_pleveldz2tmean": ("i", ["ctx", "i", "i", "p", "p", "f", "f", "i", "p", "pi", "f", "i"]),
    "mifc_kIndex": ("i", ["ctx", "i", "i", "p", "p", "p", "p", "p", "f", "f", "f", "i", "p", "pi", "f", "i"]),
    "mifc_ductingIndex": ("i", ["ctx", "i", "i", "p", "p", "f", "i", "p", "pi", "f", "i"]),
    "mifc_showalterIndex": ("i", ["ctx", "i", "i", "p", "p", "p", "f", "f", "i", "p", "pi", "f", "i"]),
    "mifc_boydenIndex": ("i", ["ctx", "i", "i", "p", "p", "p", "f", "f", "i", "p", "pi", "f", "i"]),
    "mifc_sweatIndex": ("i", ["ctx", "i", "i", "p", "p", "p", "p", "p", "p", "p", "p", "p", "pi", "f", "i"]),
    "mifc_seaSoundSpeed": ("i", ["ctx", "i", "i", "p", "p", "f", "i", "p", "pi", "f", "i"]),
    "mifc_cvtemp": ("i", ["ctx", "i", "i", "p", "i", "p", "pi", "f", "i"]),
    "mifc_abshum": ("i", ["ctx", "i", "i", "p", "p", "p", "pi", "f", "i"]),
    "mifc_windCooling": ("i", ["ctx", "i", "i", "p", "p", "p", "i", "p", "pi", "f", "i"]),
    "mifc_underCooledRain": ("i", ["ctx", "i", "i", "p", "p", "p", "f", "f", "f", "p", "pi", "f", "i"]),
    "mifc_pressure2FlightLevel": ("i", ["ctx", "i", "i", "p", "p", "pi", "f", "i"]),
    "mifc_snow_in_cm": ("i", ["ctx", "i", "i", "p", "p", "p", "p", "pi", "f", "i"]),
    "mifc_values2classes": ("i", ["ctx", "i", "i", "p", "p", "p", "i", "pi", "f", "i"]),
    "mifc_shapiro2_filter": ("i", ["ctx", "i", "i", "p", "p", "pi", "f", "i"]),
    "mifc_vesselIcingOverland": ("i", ["ctx", "i", "i", "p", "p", "p", "p", "p", "p", "p", "pi", "f", "i"]),
    "mifc_vesselIcingMertins": ("i", ["ctx", "i", "i", "p", "p", "p", "p", "p", "p", "p", "pi", "f", "i"]),
    "mifc_winddir": ("i", ["ctx", "i", "i", "p", "p", "p", "pi", "f", "i"]),
    "mifc_minvalueFields": ("i", ["ctx", "i", "i", "p", "p", "p", "pi", "f", "i"]),
    "mifc_maxvalueFields": ("i", ["ctx", "i", "i", "p", "p", "p", "pi", "f", "i"]),
    "mifc_minvalueFieldConst": ("i", ["ctx", "i", "i", "p", "f", "p", "pi", "f", "i"]),
    "mifc_maxvalueFieldConst": ("i", ["ctx", "i", "i", "p", "f", "p", "pi", "f", "i"]),
    "mifc_absvalueField": ("i", ["ctx", "i", "i", "p", "p", "pi", "f", "i"]),
    "mifc_log10Field": ("i", ["ctx", "i", "i", "p", "p", "pi", "f", "i"]),
    "mifc_pow10Field": ("i", ["ctx", "i", "i", "p", "p", "pi", "f", "i"]),
    "mifc_logField": ("i", ["ctx", "i", "i", "p", "p", "pi", "f", "i"]),
    "mifc_expField": ("i", ["ctx", "i", "i", "p", "p", "pi", "f", "i"]),
    "mifc_powerField": ("i", ["ctx", "i", "i", "p", "f", "p", "pi", "f", "i"]),
    "mifc_replaceUndefined": ("i", ["ctx", "i", "i", "p", "f", "p", "pi", "f", "i"]),
    "mifc_replaceDefined": ("i", ["ctx", "i", "i", "p", "f", "p", "pi", "f", "i"]),
    "mifc_fieldOPERfield": ("i", ["ctx", "i", "i", "i", "p", "p", "p", "pi", "f", "i"]),
    "mifc_fieldOPERconstant": ("i", ["ctx", "i", "i", "i", "p", "f", "p", "pi", "f", "i"]),
    "mifc_constantOPERfield": ("i", ["ctx", "i", "i", "i", "f", "p", "p", "pi", "f", "i"]),
    # reductions over ensemble members
    "mifc_sumFields": ("i", ["ctx", "i", "i", "p", "i", "p", "pi", "f", "i"]),
    "mifc_meanValue": ("i", ["ctx", "i", "i", "p", "p", "i", "p", "pi", "f", "i"]),
    "mifc_stddevValue": ("i", ["ctx", "i", "i", "p", "p", "i", "p", "pi", "f", "i"]),
    "mifc_extremeValue": ("i", ["ctx", "i", "i", "i", "p", "i", "p", "pi", "f", "i"]),
    "mifc_probability": ("i", ["ctx", "i", "i", "i", "p", "p", "i", "p", "i", "p", "pi", "f", "i"]),
    # batched
    "mifc_vortdiv_levels": ("i", ["ctx", "i", "i", "i", "p", "p", "p", "p", "p", "p", "pi", "f", "i"]),
    "mifc_stencil_levels": ("i", ["ctx", "i", "i", "i", "i", "p", "p", "p", "p", "p", "p", "p", "pi", "f", "i"]),
    "mifc_vortdiv_ff_levels_enqueue": ("i", ["ctx", "i", "i", "i", "p", "p", "p", "p", "p", "p", "p", "pi", "f", "pu", "pu"]),
    "mifc_stencil_levels_enqueue": ("i", ["ctx", "i", "i", "i", "i", "p", "p", "p", "p", "p", "p", "p", "pi", "f", "pu"]),
    "mifc_stencil_count_domain": ("u64", ["i", "i", "i"]),
    "mifc_last_stencil_form": ("s", []),
    "mifc_stencil_levels_ex": ("i", ["ctx", "i", "i", "i", "i", "p", "p", "p", "p", "p", "p", "p", "f", "i", "p", "p", "pi", "f", "i"]),
    "mifc_vortdiv_levels_enqueue": ("i", ["ctx", "i", "i", "i", "p", "p", "p", "p", "p", "p", "pi", "f", "pu"]),
    "mifc_batch_level_stride": ("z", ["i", "i"]),
    "mifc_vortdiv_levels_strided_enqueue": ("i", ["ctx", "i", "i", "i", "p", "p", "p", "p", "p", "p", "z", "z", "pi", "f", "pu"]),
    "mifc_hlevel_derived_levels": (
        "i",
        ["ctx", "i", "i", "i", "p", "p", "p", "p", "p", "p", "p", "p", "p", "p", "pi", "pi", "pi", "pi", "pi", "f", "i"],
    ),
    "mifc_hlevel_derived_levels_enqueue": (
        "i",
        ["ctx", "i", "i", "i", "p", "p", "p", "p", "p", "p", "p", "p", "p", "p", "pi", "pi", "f", "pu"],
    ),
    "mifc_hlevel_derived_batch": (
        "i",
        ["ctx", "i", "i", "i", "p", "p", "p", "p", "p", "p", "p", "p", "p", "s", "i", "p", "s", "i", "p", "s", "i", "p", "pi", "pi", "pi", "pi", "pi",
         "pi", "pi", "f", "i"],
    ),
    "mifc_hlevel_derived_batch_enqueue": (
        "i",
        ["ctx", "i", "i", "i", "p", "p", "p", "p", "p", "p", "p", "p", "p", "s", "i", "p", "s", "i", "p", "s", "i", "p", "pi", "pi", "f", "pu"],
    ),
    "mifc_vortdiv_slab_enqueue": ("i", ["ctx", "i", "i", "i", "i", "p", "p", "p", "p", "p", "p", "i", "f", "pu"]),
    "mifc_vortdiv_slab_rows_enqueue": ("i", ["ctx", "i", "i", "i", "i", "i", "i", "p", "p", "p", "p", "p", "p", "i", "f", "pu", "i"]),
    "mifc_halo_copy_enqueue": ("i", ["ctx", "p", "ctx", "p", "z"]),
    # a sequence of *_enqueue calls recorded into a HIP graph
    "mifc_graph_begin": ("i", ["ctx", "i"]),
    "mifc_graph_begin_lanes": ("i", ["ctx", "i", "i"]),
    "mifc_graph_lane": ("i", ["ctx", "i"]),
    "mifc_graph_end": ("p", ["ctx"]),
    "mifc_graph_launch": ("i", ["p"]),
    "mifc_graph_destroy": (None, ["p"]),
    # the decomposed step as one call (RCCL from C++, HIP-graph replay)
    "mifc_comm_unique_id": ("i", ["p"]),
    "mifc_comm_init": ("i", ["ctx", "p", "i", "i"]),
    "mifc_comm_adopt": ("i", ["ctx", "p"]),
    "mifc_comm_release": ("i", ["ctx"]),
    "mifc_comm_info": ("i", ["ctx", "pi", "pi"]),
    "mifc_slab_plan_create": ("p", ["ctx", "i", "i", "i", "i", "i", "p", "p", "p", "p", "p", "p", "i", "f", "pu"]),
    "mifc_slab_plan_destroy": (None, ["p"]),
    "mifc_slab_plan_step": ("i", ["p"]),
    "mifc_slab_plan_uses_graph": ("i", ["p"]),
    "mifc_slab_plan_begin": ("i", ["p"]),
    "mifc_slab_plan_finish": ("i", ["p"]),
}


# entry points of the measurement build only (include/mifc_measure.h): bound when the loaded library has them, i.e. when
# MIFC_LIB_PATH points a tool at the measurement build
MEASURE_SIGNATURES = {
    "mifc_timing_begin": ("i", ["ctx"]),
    "mifc_timing_end_ms": ("fl", ["ctx"]),
    "mifc_bench_stream2": ("i", ["ctx", "i", "i", "p", "p", "p", "p", "z"]),
    "mifc_diag_division": ("i", ["ctx", "p", "p", "p", "p", "p", "z"]),
}


def load_library(path=LIB_PATH):
    if not os.path.exists(path):
        raise ImportError(
            "%s not found: build the HIP extension first (python -c 'import __graft_entry__ as g; g.build()' "
            "or make -C mi-fieldcalc_amd). There is no CPU fallback." % path
        )
    # PyTorch-ROCm wheels bundle their own HIP runtime under the same SONAME
    # (libamdhip64.so.7).  A process must hold ONE runtime, or the second one
    # finds no device: when torch is importable, load it first so that
    # libmifc.so binds to the runtime torch already brought in and device
    # pointers / streams can be shared.  Plain C/C++ callers use /opt/rocm's.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    lib = ctypes.CDLL(path)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the library lacks a declared symbol
        fn.restype = _T[res] if res else None
        fn.argtypes = [_T[a] for a in args]
    for name, (res, args) in MEASURE_SIGNATURES.items():
        fn = getattr(lib, name, None)
        if fn is not None:
            fn.restype = _T[res] if res else None
            fn.argtypes = [_T[a] for a in args]
    return lib


_lib = None


def lib():
    global _lib
    if _lib is None:
        _lib = load_library()
    return _lib
