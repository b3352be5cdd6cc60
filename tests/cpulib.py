"""ctypes binding of the CPU checkers (TEST INFRASTRUCTURE).

``CpuLib("oracle")``  -> oracle/libmifc_oracle.so   (from-scratch restatement)
``CpuLib("ref")``     -> oracle/_ref/libmifc_ref.so (the real reference, compiled
                         from /root/reference by oracle/Makefile)
Both export the flat ABI of oracle/oracle_abi.h.  Nothing outside tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg may import this.
"""
import ctypes
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PATHS = {
    "oracle": (os.path.join(ROOT, "oracle", "libmifc_oracle.so"), "mifcorc_"),
    "ref": (os.path.join(ROOT, "oracle", "_ref", "libmifc_ref.so"), "mifcref_"),
    "ref_omp": (os.path.join(ROOT, "oracle", "_ref", "libmifc_ref_omp.so"), "mifcref_"),
}

ALL_DEFINED, NONE_DEFINED, SOME_DEFINED = 0, 1, 2
UNDEF = np.float32(1.0e35)

_F = ctypes.c_void_p
_I = ctypes.c_int
_R = ctypes.c_float
_S = ctypes.c_char_p

# name -> argtypes after (nx, ny): 'p' field pointer, 'f' float, 's' string, 'i' int, 'o' output pointer,
# 'T' table of field pointers (argument: list of arrays), 'D' int array, one per field of the table
# (argument: list of flags), 'n' the table's length (filled in here), 'V' float array + its length
# (argument: list of floats).  A leading 'C' = an int `compute` BEFORE (nx, ny), as the reference has it.
SIGS = {
    "vectorabs": "ppo",
    "relvort": "ppppo",
    "absvort": "pppppo",
    "divergence": "ppppo",
    "gradient": "pppio",
    "plevelgwind_xcomp": "ppppo",
    "plevelgwind_ycomp": "ppppo",
    "plevelgvort": "ppppo",
    "ilevelgwind": "ppppoo",
    "pleveltemp": "pfsio",
    "hleveltemp": "ppffsio",
    "aleveltemp": "ppsio",
    "plevelhum": "ppfsio",
    "hlevelhum": "pppffsio",
    "alevelhum": "pppsio",
    "cvhum": "ppsio",
    # SURVEY.md 8f-1
    "advection": "pppppfo",
    "jacobian": "ppppo",
    "momentumXcoordinate": "pppfo",
    "momentumYcoordinate": "pppfo",
    "thermalFrontParameter": "pppo",
    "plevelqvector": "pppppfio",
    # SURVEY.md 8f-3
    "plevelthe": "ppfio",
    "hlevelthe": "pppffio",
    "alevelthe": "pppio",
    "plevelducting": "ppfio",
    "hlevelducting": "pppffio",
    "alevelducting": "pppio",
    "hlevelpressure": "pffo",
    "pleveldz2tmean": "ppffio",
    "kIndex": "pppppfffio",
    "ductingIndex": "ppfio",
    "showalterIndex": "pppffio",
    "boydenIndex": "pppffio",
    "sweatIndex": "ppppppppo",
    "seaSoundSpeed": "ppfio",
    "cvtemp": "pio",
    "abshum": "ppo",
    "windCooling": "pppio",
    "underCooledRain": "pppfffo",
    "pressure2FlightLevel": "po",
    "snow_in_cm": "pppo",
    "values2classes": "poV",
    "shapiro2_filter": "po",
    "vesselIcingOverland": "ppppppo",
    "vesselIcingMertins": "ppppppo",
    "minvalueFields": "ppo",
    "maxvalueFields": "ppo",
    "minvalueFieldConst": "pfo",
    "maxvalueFieldConst": "pfo",
    "absvalueField": "po",
    "log10Field": "po",
    "pow10Field": "po",
    "logField": "po",
    "expField": "po",
    "powerField": "pfo",
    "replaceUndefined": "pfo",
    "replaceDefined": "pfo",
    "fieldOPERfield": "Cppo",
    "fieldOPERconstant": "Cpfo",
    "constantOPERfield": "Cfpo",
    # SURVEY.md 8f-4
    "sumFields": "Tno",
    "meanValue": "TDno",
    "stddevValue": "TDno",
    "extremeValue": "CTno",
    "probability": "CTDnVo",
}
# restatement only: the product's EXTENSION "wind direction from u/v" has no reference function
ORACLE_ONLY = {"winddir": "ppo"}
_CT = {"p": [_F], "o": [_F], "f": [_R], "s": [_S], "i": [_I], "T": [_F], "D": [_F], "n": [_I], "V": [_F, _I]}


def available(which):
    return os.path.exists(PATHS[which][0])


class CpuLib:
    def __init__(self, which="oracle"):
        path, prefix = PATHS[which]
        if not os.path.exists(path):
            raise FileNotFoundError(path + " (run: make -C oracle)")
        self.which = which
        self._lib = ctypes.CDLL(path)
        self._fn = {}
        self._sigs = dict(SIGS, **(ORACLE_ONLY if which == "oracle" else {}))
        for name, sig in self._sigs.items():
            fn = getattr(self._lib, prefix + name)
            fn.restype = _I
            lead = [_I] if sig.startswith("C") else []
            fn.argtypes = lead + [_I, _I] + [t for c in sig.lstrip("C") for t in _CT[c]] + [ctypes.c_void_p, _R]
            self._fn[name] = fn
        kind = getattr(self._lib, prefix + "kind")
        kind.restype = _S
        self.kind = kind().decode()

    def call(self, name, nx, ny, *args, fdefined=SOME_DEFINED, undef=UNDEF, outs=None):
        """args in reference order (fields as numpy float32 arrays, scalars,
        strings); outputs are allocated here.  Returns (ok, out or (out0,out1), flag)."""
        sig = self._sigs[name]
        lead = []
        if sig.startswith("C"):
            lead, args, sig = [int(args[0])], args[1:], sig[1:]
        n_in = sum(1 for c in sig if c not in "on")
        assert len(args) == n_in, (name, len(args), n_in)
        cargs, keep = [], []
        it = iter(args)
        n_out = sig.count("o")
        n_table = 0
        if outs is None:
            outs = [np.empty((ny, nx), dtype=np.float32) for _ in range(n_out)]
        oi = iter(outs)
        for c in sig:
            if c == "p":
                a = next(it)
                if a is None:
                    cargs.append(None)
                else:
                    a = np.ascontiguousarray(a, dtype=np.float32)
                    keep.append(a)
                    cargs.append(a.ctypes.data)
            elif c == "o":
                cargs.append(next(oi).ctypes.data)
            elif c == "T":
                fields = [np.ascontiguousarray(a, dtype=np.float32) for a in next(it)]
                table = (ctypes.c_void_p * max(len(fields), 1))(*[a.ctypes.data for a in fields])
                keep += fields + [table]
                n_table = len(fields)
                cargs.append(ctypes.addressof(table))
            elif c == "D":
                flags = (ctypes.c_int * max(n_table, 1))(*[int(x) for x in next(it)])
                keep.append(flags)
                cargs.append(ctypes.addressof(flags))
            elif c == "n":
                cargs.append(n_table)
            elif c == "V":
                vals = np.ascontiguousarray(next(it), dtype=np.float32)
                keep.append(vals)
                cargs += [vals.ctypes.data, int(vals.size)]
            elif c == "s":
                cargs.append(next(it).encode())
            elif c == "f":
                cargs.append(float(next(it)))
            else:
                cargs.append(int(next(it)))
        fd = ctypes.c_int(int(fdefined))
        ok = self._fn[name](*lead, nx, ny, *cargs, ctypes.addressof(fd), float(undef))
        res = outs[0] if n_out == 1 else tuple(outs)
        return bool(ok), res, fd.value

    def raw(self, name):
        return self._fn[name]
