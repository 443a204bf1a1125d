// DEV-ONLY: relative issue cost of VALU instruction classes on gfx950 (wave64), to price the kernel's mix.
//   hipcc -O3 --offload-arch=gfx950 -o scripts/_variants/ubench_valu.bin scripts/ubench_valu.hip   (here; it travels with gpurun)
//   scripts/_variants/ubench_valu.bin                                                           (on the GPU box)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
constexpr int ITER = 2048, UNR = 8;
typedef float f2 __attribute__((ext_vector_type(2)));
template <int KIND> __global__ void __launch_bounds__(256) k(float* out, float seed)
{
    float a[UNR]; double d[UNR]; f2 p[UNR];
    for (int j = 0; j < UNR; ++j) { a[j] = seed + j + threadIdx.x; d[j] = a[j]; p[j] = f2{a[j], a[j] + 1.0f}; }
    const float m = seed * 0.999f, c = seed * 1e-3f; const double md = m, cd = c; const f2 mp{m, m}, cp{c, c};
    for (int i = 0; i < ITER; ++i) {
#pragma unroll
        for (int j = 0; j < UNR; ++j) {
            if constexpr (KIND == 0) a[j] = __builtin_fmaf(a[j], m, c);
            if constexpr (KIND == 1) d[j] = __builtin_fma(d[j], md, cd);
            if constexpr (KIND == 2) p[j] = __builtin_elementwise_fma(p[j], mp, cp);
            if constexpr (KIND == 3) d[j] = d[j] * md;
            if constexpr (KIND == 4) d[j] = d[j] + cd;
            if constexpr (KIND == 5) { a[j] = (float)((double)a[j] * md); }                 // cvt + mul_f64 + cvt
            if constexpr (KIND == 6) a[j] = __builtin_amdgcn_rcpf(a[j]) + c;
            if constexpr (KIND == 7) a[j] = (a[j] > c) ? a[j] * m : c;                        // cmp + cndmask + mul
            if constexpr (KIND == 8) { a[j] = __builtin_fmaf(a[j], m, c); d[j] = __builtin_fma(d[j], md, cd); }   // alternating fp32 / fp64
            if constexpr (KIND == 9) { a[j] = __builtin_fmaf(a[j], m, c); p[j] = __builtin_elementwise_fma(p[j], mp, cp); }
        }
    }
    float s = 0; for (int j = 0; j < UNR; ++j) s += a[j] + (float)d[j] + p[j].x + p[j].y;
    if (s == 12345.678f) out[0] = s;
}
template <int KIND> float run(const char* name, int instr_per_iter)
{
    float* out; hipMalloc(&out, 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int blocks = 256 * 8;                         // 8 blocks of 4 waves per CU: 8 waves per SIMD
    hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(256), 0, 0, out, 1.0f);
    hipDeviceSynchronize();
    hipEventRecord(e0); for (int r = 0; r < 5; ++r) hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(256), 0, 0, out, 1.0f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
    const double wave_instr_per_simd = 8.0 * ITER * UNR * instr_per_iter;     // 8 waves per SIMD
    printf("%-28s %8.3f ms   %6.2f ns per wave-instruction per SIMD  (= %.2f cycles at 2.4 GHz)\n", name, ms,
           ms * 1e6 / wave_instr_per_simd, ms * 1e6 / wave_instr_per_simd * 2.4);
    hipFree(out); return ms;
}
int main()
{
    run<0>("v_fma_f32", 1); run<1>("v_fma_f64", 1); run<2>("v_pk_fma_f32", 1); run<3>("v_mul_f64", 1); run<4>("v_add_f64", 1);
    run<5>("cvt+mul_f64+cvt (3 instr)", 3); run<6>("v_rcp_f32 + add (2 instr)", 2); run<7>("cmp+cndmask+mul (3 instr)", 3);
    run<8>("fma_f32 + fma_f64 alternating (2)", 2); run<9>("fma_f32 + pk_fma_f32 alternating (2)", 2);
    return 0;
}
