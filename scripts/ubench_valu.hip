// DEV-ONLY: relative issue cost of VALU instruction classes on gfx950 (wave64), to price the kernel's mix.
//   hipcc -O3 --offload-arch=gfx950 -o scripts/_variants/ubench_valu.bin scripts/ubench_valu.hip   (here; it travels with gpurun)
//   scripts/_variants/ubench_valu.bin                                                           (on the GPU box)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
constexpr int ITER = 2048, UNR = 8;
typedef float f2 __attribute__((ext_vector_type(2)));
template <int KIND> __global__ void __launch_bounds__(256) k(float* out, float seed, uint64_t* clk)
{
    float a[UNR]; double d[UNR]; f2 p[UNR]; uint32_t u[UNR];
    for (int j = 0; j < UNR; ++j) { a[j] = seed + j + threadIdx.x; d[j] = a[j]; p[j] = f2{a[j], a[j] + 1.0f}; u[j] = threadIdx.x * 2654435761u + j; }
    const uint64_t t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
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
            if constexpr (KIND == 10) { d[j] = (double)a[j] + d[j]; a[j] = a[j] * m; }            // cvt_f64_f32 + add_f64 + mul_f32
            if constexpr (KIND == 11) { a[j] = (float)d[j] + a[j]; d[j] = d[j] + cd; }            // cvt_f32_f64 + add_f32 + add_f64
            if constexpr (KIND == 12) { u[j] = __builtin_amdgcn_alignbit(u[j], (uint32_t)(__double_as_longlong(d[j]) >> 32), 31); d[j] = d[j] + cd; }   // alignbit + add_f64
            if constexpr (KIND == 13) { d[j] = (d[j] > cd) ? d[j] * md : cd; }                     // cmp_f64 + 2 cndmask + mul_f64
            if constexpr (KIND == 14) { a[j] = a[j] * m; }                                         // v_mul_f32
            if constexpr (KIND == 15) { u[j] = __builtin_popcount(u[j] & 0x5a5a5a5au) + u[j]; }      // and + bcnt(+add folded)
            if constexpr (KIND == 16) { d[j] = __builtin_fmax(d[j] * md, cd); }                     // mul_f64 + max_f64
        }
    }
    const uint64_t t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    float s = 0; for (int j = 0; j < UNR; ++j) s += a[j] + (float)d[j] + p[j].x + p[j].y + (float)u[j];
    if (s == 12345.678f) out[0] = s;
    if (blockIdx.x == 0 && threadIdx.x == 0) { clk[0] = t1 - t0; clk[1] = r1 - r0; }
}
template <int KIND, int WAVES = 8> float run(const char* name, int instr_per_iter)
{
    float* out; hipMalloc(&out, 4);
    uint64_t* clk; hipMalloc(&clk, 16); uint64_t hclk[2] = {0, 0};
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int blocks = 256 * WAVES;                     // WAVES blocks of 4 waves per CU: WAVES waves per SIMD
    hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(256), 0, 0, out, 1.0f, clk);
    hipDeviceSynchronize();
    hipEventRecord(e0); for (int r = 0; r < 5; ++r) hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(256), 0, 0, out, 1.0f, clk);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
    hipMemcpy(hclk, clk, 16, hipMemcpyDeviceToHost);
    const double ghz = hclk[1] ? (double)hclk[0] / (double)hclk[1] * 0.1 : 0.0;          // s_memrealtime ticks at 100 MHz
    const double wave_instr_per_simd = (double)WAVES * ITER * UNR * instr_per_iter;
    printf("%-36s %d waves/SIMD %8.3f ms   %6.2f ns per wave-instruction per SIMD = %5.2f cycles at the measured %.2f GHz (in-kernel: %.2f cycles per wave-instruction of ONE wave)\n",
           name, WAVES, ms, ms * 1e6 / wave_instr_per_simd, ms * 1e6 / wave_instr_per_simd * ghz, ghz,
           (double)hclk[0] / ((double)ITER * UNR * instr_per_iter));
    hipFree(out); hipFree(clk); return ms;
}
int main()
{
    run<0>("v_fma_f32", 1); run<1>("v_fma_f64", 1); run<2>("v_pk_fma_f32", 1); run<3>("v_mul_f64", 1); run<4>("v_add_f64", 1);
    run<5>("cvt+mul_f64+cvt (3 instr)", 3); run<6>("v_rcp_f32 + add (2 instr)", 2); run<7>("cmp+cndmask+mul (3 instr)", 3);
    run<8>("fma_f32 + fma_f64 alternating (2)", 2); run<9>("fma_f32 + pk_fma_f32 alternating (2)", 2);
    run<10>("cvt_f64_f32 + add_f64 + mul_f32 (3)", 3); run<11>("cvt_f32_f64 + add_f32 + add_f64 (3)", 3);
    run<12>("alignbit + add_f64 (2)", 2); run<13>("cmp_f64 + 2 cndmask + mul_f64 (4)", 4); run<14>("v_mul_f32", 1);
    run<15>("and + bcnt (2)", 2); run<16>("mul_f64 + max_f64 (2)", 2);
    printf("-- 4 waves per SIMD (the wrench kernels' occupancy)\n");
    run<0, 4>("v_fma_f32", 1); run<1, 4>("v_fma_f64", 1); run<3, 4>("v_mul_f64", 1); run<4, 4>("v_add_f64", 1);
    run<10, 4>("cvt_f64_f32 + add_f64 + mul_f32 (3)", 3); run<13, 4>("cmp_f64 + 2 cndmask + mul_f64 (4)", 4);
    printf("-- 1 wave per SIMD\n");
    run<0, 1>("v_fma_f32", 1); run<1, 1>("v_fma_f64", 1);
    return 0;
}
