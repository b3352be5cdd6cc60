// valu_rate_probe.hip -- measurement only (tools/): issue rate of the VALU instructions the double-precision "combine"
// steps of the kernels are made of (conversions f32 <-> f64, f64 add / mul / fma / ldexp, the f64 reciprocal), relative
// to v_add_f32, on gfx950.  Every SIMD runs 4 waves; each wave issues 8 independent chains of the instruction, UNROLL
// times per loop trip; cycles per wave-instruction = time x clock / (instructions issued per SIMD).
//
// build: hipcc --offload-arch=gfx950 -O2 -o tools/valu_rate_probe tools/valu_rate_probe.hip
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

#define CHECK(x)                                                                \
  do {                                                                          \
    hipError_t e_ = (x);                                                        \
    if (e_ != hipSuccess) {                                                     \
      std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));             \
      std::exit(1);                                                             \
    }                                                                           \
  } while (0)

constexpr int TRIPS = 2000;
constexpr int CHAINS = 8;
constexpr int UNROLL = 8;

enum Op { ADD_F32, PK_ADD_F32, ADD_F64, MUL_F64, FMA_F64, LDEXP_F64, CVT_F64_F32, CVT_F32_F64, RCP_F64, SQRT_F32, RCP_F32, CNDMASK, FMA_F32, DIV_FIXUP_F64,
          CNDMASK_SGPR, CNDMASK_3REG, CMP_VCC, CMP_SGPR, CMP_THEN_CNDMASK, MOV, MOV_DPP, BFI, MAX_F32, PK_MUL_F32, PK_FMA_F32, CMP_CLASS };

template <int OP>
__global__ __launch_bounds__(256) void probe(float* out, float seed)
{
  float f[CHAINS];
  double d[CHAINS];
  for (int k = 0; k < CHAINS; ++k) {
    f[k] = seed + (float)(threadIdx.x + k);
    d[k] = (double)seed + (double)(threadIdx.x * 3 + k);
  }
  for (int t = 0; t < TRIPS; ++t) {
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) {
#pragma unroll
      for (int k = 0; k < CHAINS; ++k) {
        if (OP == ADD_F32)
          asm volatile("v_add_f32 %0, %0, %0" : "+v"(f[k]));
        else if (OP == FMA_F32)
          asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(f[k]));
        else if (OP == PK_ADD_F32)
          asm volatile("v_pk_add_f32 %0, %0, %0" : "+v"(d[k]));
        else if (OP == ADD_F64)
          asm volatile("v_add_f64 %0, %0, %0" : "+v"(d[k]));
        else if (OP == MUL_F64)
          asm volatile("v_mul_f64 %0, %0, %0" : "+v"(d[k]));
        else if (OP == FMA_F64)
          asm volatile("v_fma_f64 %0, %0, %0, %0" : "+v"(d[k]));
        else if (OP == LDEXP_F64)
          asm volatile("v_ldexp_f64 %0, %0, 1" : "+v"(d[k]));
        else if (OP == CVT_F64_F32)
          asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(d[k]) : "v"(f[k]));
        else if (OP == CVT_F32_F64)
          asm volatile("v_cvt_f32_f64 %0, %1" : "=v"(f[k]) : "v"(d[k]));
        else if (OP == RCP_F64)
          asm volatile("v_rcp_f64 %0, %0" : "+v"(d[k]));
        else if (OP == SQRT_F32)
          asm volatile("v_sqrt_f32 %0, %0" : "+v"(f[k]));
        else if (OP == RCP_F32)
          asm volatile("v_rcp_f32 %0, %0" : "+v"(f[k]));
        else if (OP == CNDMASK)
          asm volatile("v_cndmask_b32 %0, %0, %0, vcc" : "+v"(f[k]));
        else if (OP == DIV_FIXUP_F64)
          asm volatile("v_div_fixup_f64 %0, %0, %0, %0" : "+v"(d[k]));
        else if (OP == CNDMASK_SGPR)
          asm volatile("v_cndmask_b32_e64 %0, %0, %1, s[20:21]" : "+v"(f[k]) : "v"(f[(k + 1) % CHAINS]) : "s20", "s21");
        else if (OP == CNDMASK_3REG)
          asm volatile("v_cndmask_b32 %0, %1, %2, vcc" : "=v"(f[k]) : "v"(f[(k + 1) % CHAINS]), "v"(f[(k + 2) % CHAINS]) : "vcc");
        else if (OP == CMP_VCC)
          asm volatile("v_cmp_gt_f32 vcc, %0, %1" : : "v"(f[k]), "v"(f[(k + 1) % CHAINS]) : "vcc");
        else if (OP == CMP_SGPR)
          asm volatile("v_cmp_gt_f32_e64 s[20:21], %0, %1" : : "v"(f[k]), "v"(f[(k + 1) % CHAINS]) : "s20", "s21");
        else if (OP == CMP_THEN_CNDMASK)
          asm volatile("v_cmp_gt_f32 vcc, %0, %1\n\tv_cndmask_b32 %0, %0, %1, vcc" : "+v"(f[k]) : "v"(f[(k + 1) % CHAINS]) : "vcc");
        else if (OP == MOV)
          asm volatile("v_mov_b32 %0, %1" : "=v"(f[k]) : "v"(f[(k + 1) % CHAINS]));
        else if (OP == MOV_DPP)
          asm volatile("v_mov_b32_dpp %0, %1 wave_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(f[k]) : "v"(f[(k + 1) % CHAINS]));
        else if (OP == BFI)
          asm volatile("v_bfi_b32 %0, %1, %0, %2" : "+v"(f[k]) : "v"(f[(k + 1) % CHAINS]), "v"(f[(k + 2) % CHAINS]));
        else if (OP == MAX_F32)
          asm volatile("v_max_f32 %0, %0, %1" : "+v"(f[k]) : "v"(f[(k + 1) % CHAINS]));
        else if (OP == PK_MUL_F32)
          asm volatile("v_pk_mul_f32 %0, %0, %0" : "+v"(d[k]));
        else if (OP == PK_FMA_F32)
          asm volatile("v_pk_fma_f32 %0, %0, %0, %0" : "+v"(d[k]));
        else if (OP == CMP_CLASS)
          asm volatile("v_cmp_class_f32 vcc, %0, %1" : : "v"(f[k]), "v"(f[(k + 1) % CHAINS]) : "vcc");
      }
    }
  }
  float acc = 0.f;
  for (int k = 0; k < CHAINS; ++k)
    acc += f[k] + (float)d[k];
  if (acc == 123.456f)
    out[0] = acc;
}

template <int OP>
static double run(const char* name, float* out, double base_ns)
{
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0));
  CHECK(hipEventCreate(&e1));
  const int blocks = 256 * 4; // 4 workgroups of 4 waves per CU: 4 waves per SIMD
  hipLaunchKernelGGL((probe<OP>), dim3(blocks), dim3(256), 0, 0, out, 1.0f);
  CHECK(hipDeviceSynchronize());
  float best = 1e30f;
  for (int rep = 0; rep < 5; ++rep) {
    CHECK(hipEventRecord(e0, 0));
    hipLaunchKernelGGL((probe<OP>), dim3(blocks), dim3(256), 0, 0, out, 1.0f);
    CHECK(hipEventRecord(e1, 0));
    CHECK(hipEventSynchronize(e1));
    float ms;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    if (ms < best)
      best = ms;
  }
  const double per_simd = 4.0 * TRIPS * UNROLL * CHAINS; // wave-instructions per SIMD
  const double ns = best * 1e6 / per_simd;
  std::printf("%-16s %8.3f ns per wave-instruction per SIMD  = %5.2f x v_add_f32\n", name, ns, base_ns > 0 ? ns / base_ns : 1.0);
  return ns;
}

int main()
{
  float* out;
  CHECK(hipMalloc(&out, 64));
  const double base = run<ADD_F32>("v_add_f32", out, 0);
  run<FMA_F32>("v_fma_f32", out, base);
  run<PK_ADD_F32>("v_pk_add_f32", out, base);
  run<CNDMASK>("v_cndmask_b32", out, base);
  run<ADD_F64>("v_add_f64", out, base);
  run<MUL_F64>("v_mul_f64", out, base);
  run<FMA_F64>("v_fma_f64", out, base);
  run<LDEXP_F64>("v_ldexp_f64", out, base);
  run<DIV_FIXUP_F64>("v_div_fixup_f64", out, base);
  run<CVT_F64_F32>("v_cvt_f64_f32", out, base);
  run<CVT_F32_F64>("v_cvt_f32_f64", out, base);
  run<CNDMASK_3REG>("cndmask 3 regs", out, base);
  run<CNDMASK_SGPR>("cndmask sgpr", out, base);
  run<CMP_VCC>("v_cmp vcc", out, base);
  run<CMP_SGPR>("v_cmp sgpr pair", out, base);
  run<CMP_CLASS>("v_cmp_class vcc", out, base);
  run<CMP_THEN_CNDMASK>("cmp+cndmask (2)", out, base);
  run<MOV>("v_mov_b32", out, base);
  run<MOV_DPP>("v_mov_b32_dpp", out, base);
  run<BFI>("v_bfi_b32", out, base);
  run<MAX_F32>("v_max_f32", out, base);
  run<PK_MUL_F32>("v_pk_mul_f32", out, base);
  run<PK_FMA_F32>("v_pk_fma_f32", out, base);
  run<RCP_F64>("v_rcp_f64", out, base);
  run<RCP_F32>("v_rcp_f32", out, base);
  run<SQRT_F32>("v_sqrt_f32", out, base);
  return 0;
}
