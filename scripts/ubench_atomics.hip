// DEV: what one device-scope ticket / fence costs per block on gfx950 (why the kinetic-energy reduction is shaped the way
// it is: DESIGN.md).  N blocks of 256 threads, each does ONE of the operations below; time per launch and per block.
//   hipcc -O3 --offload-arch=gfx950 -o scripts/_variants/ubench_atomics scripts/ubench_atomics.hip && scripts/_variants/ubench_atomics
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>

template <int MODE>
__global__ void __launch_bounds__(256) k(uint32_t* c, double* p)
{
    if constexpr (MODE == 0) { if (threadIdx.x == 0 && c == nullptr) p[0] = 1.0; }
    if constexpr (MODE == 1) { if (threadIdx.x == 0) __hip_atomic_fetch_add(c, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
    if constexpr (MODE == 2) { if (threadIdx.x == 0) __hip_atomic_fetch_add(c, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT); }
    if constexpr (MODE == 3) { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent"); }
    if constexpr (MODE == 4) { if (threadIdx.x == 0) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent"); }
    if constexpr (MODE == 5) { if (threadIdx.x == 0) __hip_atomic_fetch_add(c, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
    if constexpr (MODE == 6) { if (threadIdx.x == 0) __hip_atomic_fetch_add(c + 64 * blockIdx.x, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
    if constexpr (MODE == 7) { if (threadIdx.x == 0) __hip_atomic_fetch_add(c, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }
    if constexpr (MODE == 8) { if (threadIdx.x == 0) __hip_atomic_fetch_add(c + 64 * (blockIdx.x & 31u), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
    if constexpr (MODE == 9) {   // the pattern of ke_finish_block: write-through partial, release fence, barrier, acq_rel ticket
        if ((threadIdx.x & 63u) == 0) __hip_atomic_store(p + blockIdx.x * 4 + (threadIdx.x >> 6), 1.0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        __syncthreads();
        if (threadIdx.x == 0) __hip_atomic_fetch_add(c, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
    }
    if constexpr (MODE == 10) {  // no-return atomic (fire and forget)
        if (threadIdx.x == 0) __hip_atomic_fetch_add(c, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT), (void)0;
    }
    if constexpr (MODE == 11) {  // write-through store + s_waitcnt, relaxed ticket: no cache-maintenance instruction at all
        if ((threadIdx.x & 63u) == 0) __hip_atomic_store(p + blockIdx.x * 4 + (threadIdx.x >> 6), 1.0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __builtin_amdgcn_s_waitcnt(0);
        __syncthreads();
        if (threadIdx.x == 0) __hip_atomic_fetch_add(c, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

// dependent chains in ONE wavefront: what one device-scope round trip costs (ns per operation)
template <int OP>
__global__ void __launch_bounds__(64) chain(uint32_t* c, double* p, int iters, uint32_t* sink)
{
    uint32_t x = threadIdx.x == 0 ? 0u : 1u;
    for (int i = 0; i < iters; ++i) {
        if constexpr (OP == 0) { x += __hip_atomic_fetch_add(c + (x & 1u), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }                   // RMW, result needed
        if constexpr (OP == 1) { __hip_atomic_store(p + (x & 1u), 1.0 * i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); __builtin_amdgcn_s_waitcnt(0); x += 2; }   // sc1 store + wait
        if constexpr (OP == 2) { x += (uint32_t)__hip_atomic_load(p + (x & 7u), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }                   // sc1 load, result needed
        if constexpr (OP == 3) { x += (uint32_t)p[x & 7u]; __builtin_amdgcn_s_waitcnt(0); }                                                     // ordinary load (cache hit)
        if constexpr (OP == 4) { p[64 + (x & 1u)] = 1.0 * i; __builtin_amdgcn_s_waitcnt(0); x += 2; }                                            // ordinary store + wait
        if constexpr (OP == 5) { __hip_atomic_store(p + (x & 1u), 1.0 * i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent"); x += 2; }
    }
    if (x == 0xdeadbeefu) *sink = x;
}
template <int OP>
void run_chain(const char* what, uint32_t* c, double* p)
{
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 2000;
    hipLaunchKernelGGL(chain<OP>, dim3(1), dim3(64), 0, 0, c, p, 10, c + 1024);
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL(chain<OP>, dim3(1), dim3(64), 0, 0, c, p, iters, c + 1024);
    hipEventRecord(e1, 0); hipEventSynchronize(e1);
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    printf("chain: %-70s %8.1f ns per operation\n", what, ms * 1e6 / iters);
    fflush(stdout);
}

template <int MODE>
void run(const char* what, int n, uint32_t* c, double* p)
{
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 5; ++i) hipLaunchKernelGGL(k<MODE>, dim3(n), dim3(256), 0, 0, c, p);
    hipEventRecord(e0, 0);
    const int reps = 20;
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(k<MODE>, dim3(n), dim3(256), 0, 0, c, p);
    hipEventRecord(e1, 0); hipEventSynchronize(e1);
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    printf("N=%5d %-70s %8.2f us per launch  %7.1f ns per block\n", n, what, ms * 1e3 / reps, ms * 1e6 / reps / n);
    fflush(stdout);
}

int main()
{
    uint32_t* c; double* p;
    hipMalloc(&c, 1 << 22); hipMemset(c, 0, 1 << 22);
    hipMalloc(&p, 1 << 20);
    run_chain<0>("device-scope atomic add, result needed", c, p);
    run_chain<1>("write-through (sc1) store + s_waitcnt vmcnt(0)", c, p);
    run_chain<2>("device-scope (sc1) load, result needed", c, p);
    run_chain<3>("ordinary load, same lines (cache hit)", c, p);
    run_chain<4>("ordinary store + s_waitcnt vmcnt(0)", c, p);
    run_chain<5>("write-through store + release fence at device scope", c, p);
    for (int n : {256, 1024, 4096}) {
        run<0>("empty", n, c, p);
        run<1>("thread 0: relaxed agent-scope atomic add, one counter", n, c, p);
        run<10>("thread 0: the same, result unused", n, c, p);
        run<2>("thread 0: acq_rel agent-scope atomic add, one counter", n, c, p);
        run<3>("every wave: release fence at agent scope", n, c, p);
        run<4>("thread 0: release fence at agent scope", n, c, p);
        run<5>("thread 0: relaxed workgroup-scope atomic add, one counter", n, c, p);
        run<6>("thread 0: relaxed agent-scope atomic add, one counter PER BLOCK", n, c, p);
        run<7>("thread 0: relaxed system-scope atomic add, one counter", n, c, p);
        run<8>("thread 0: relaxed agent-scope atomic add, 32 counters", n, c, p);
        run<9>("ke_finish_block's pattern (wt store, fence, barrier, acq_rel ticket)", n, c, p);
        run<11>("wt store, s_waitcnt, barrier, relaxed ticket (no cache maintenance)", n, c, p);
    }
    return 0;
}
