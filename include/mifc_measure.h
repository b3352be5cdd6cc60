/*
 * mifc_measure.h -- entry points of libmifc_measure.so ONLY (the measurement build, tools/): bandwidth yardsticks, an
 * arithmetic self-check and per-launch event timing.  None of them is part of the product ABI (include/mifc.h) and
 * libmifc.so exports none of them.
 */
#ifndef MIFC_MEASURE_H
#define MIFC_MEASURE_H

#include "mifc.h"

#ifdef __cplusplus
extern "C" {
#endif

/* Measurement aid (tools/bench_ops.py): kernel time of the calls made between begin and end,
 * from HIP events recorded around every launch on the launch stream.  end returns milliseconds,
 * -1 on error or after more than 16 launches. */
int mifc_timing_begin(mifc_ctx* ctx);
float mifc_timing_end_ms(mifc_ctx* ctx);
/* Bandwidth yardstick: copies src0 -> dst0 and src1 -> dst1 (n_floats each,
 * device pointers, 16-byte aligned, n_floats % 4 == 0) with the operators'
 * access shape and no arithmetic.  variant 0: plain loads/stores, 1: nontemporal
 * stores, 2: nontemporal loads and stores.  blocks <= 0: one lane per 16 bytes;
 * otherwise a grid-stride loop over `blocks` workgroups of 256.  variant 3:
 * split-role copy (waves either load or store); variant 4: `blocks` workgroups
 * of 384 lanes, each streaming one contiguous chunk front to back; variants 5-7:
 * write-only (the sources are not read): linear, 4-row x 256-column tiles of
 * 1440-column rows, waves looping over 8 rows of such a segment; variant 8: 6-wave
 * workgroups writing full rows, 8 rows each.  Asynchronous. */
int mifc_bench_stream2(mifc_ctx* ctx, int variant, int blocks, float* dst0, float* dst1, const float* src0, const float* src1,
                       size_t n_floats);

/* Arithmetic self-check of the fused stencil kernels' division: for i < n writes
 * (float)((0.5 * a[i] * b[i] * 9.8f) / g[i]) (operands promoted to double) once
 * through the shared-reciprocal quotient the kernels use and once through a plain
 * double division; the two must agree bit for bit.  Device pointers, asynchronous. */
int mifc_diag_division(mifc_ctx* ctx, const float* a, const float* b, const float* g, float* shared, float* plain, size_t n);

#ifdef __cplusplus
} /* extern "C" */
#endif

#endif /* MIFC_MEASURE_H */
