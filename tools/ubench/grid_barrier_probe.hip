// What does a chip-wide dependency cost on gfx950?  (round 6, VERDICT r5 item 4: the 20^3 minibatch is ~38 dependent launches of 4-8 us.)
//   (1) a hipGraph chain of N small kernels on one stream (what the minibatch graph is made of): us per node, null kernels and kernels
//       that hand 512 KB from one node to the next;
//   (2) ONE persistent kernel, one workgroup per CU (or per two), with N grid barriers in between the same hand-overs: a monotonic
//       arrival counter (agent-scope atomic add, spin on an agent-scope load) with a release fence in front and an acquire fence behind.
// Every hand-over is CHECKED (workgroup b reads what workgroup b + 37 wrote in the previous phase), so a barrier that is fast because it
// is wrong shows.  Spins are bounded: a lost arrival sets an error flag instead of hanging the box.
//       hipcc --offload-arch=gfx950 -O3 -o tools/ubench/bin/grid_barrier_probe tools/ubench/grid_barrier_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

constexpr int kWordsPerWg = 1024;  // 4 KB per workgroup and phase

__global__ void k_null(int *p) { if (p == nullptr && threadIdx.x == 12345) *p = 0; }

// one phase as its own kernel: read the slot of workgroup b + 37 written by the previous phase, check, write my slot for the next one
__global__ void k_phase(const unsigned *__restrict__ src, unsigned *__restrict__ dst, unsigned phase, unsigned *__restrict__ errors)
{
    const unsigned n = gridDim.x, b = blockIdx.x, o = (b + 37) % n;
    unsigned bad = 0;
    for (int i = threadIdx.x; i < kWordsPerWg; i += blockDim.x) {
        if (phase > 0) bad += src[o * kWordsPerWg + i] != ((phase - 1) << 20 | o << 10 | (unsigned)i);
        dst[b * kWordsPerWg + i] = phase << 20 | b << 10 | (unsigned)i;
    }
    if (bad) atomicAdd(errors, bad);
}

template <int MODE>  // 0: release + acquire fences (correct);  1: relaxed atomics only (lower bound, hand-over through sc1 accesses);  2: no barrier at all (the phases' own cost)
__device__ __forceinline__ void grid_barrier(unsigned *ctr, unsigned target, unsigned *errors)
{
    if (MODE == 2) { __syncthreads(); return; }
    __syncthreads();
    if (threadIdx.x == 0) {
        if (MODE == 0) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        unsigned spins = 0;
        while (__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
            __builtin_amdgcn_s_sleep(1);
            if (++spins > (1u << 22)) { atomicAdd(errors, 1u << 20); break; }
        }
        if (MODE == 0) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    __syncthreads();
}

template <int MODE>
__global__ void k_persistent(unsigned *__restrict__ bufA, unsigned *__restrict__ bufB, int phases, unsigned *__restrict__ ctr, unsigned *__restrict__ errors)
{
    const unsigned n = gridDim.x, b = blockIdx.x, o = (b + 37) % n;
    unsigned bad = 0;
    for (int p = 0; p < phases; ++p) {
        const unsigned *src = (p & 1) ? bufA : bufB;
        unsigned *dst = (p & 1) ? bufB : bufA;
        for (int i = threadIdx.x; i < kWordsPerWg; i += blockDim.x) {
            if (p > 0 && MODE != 2) {
                unsigned v;
                if (MODE == 1) v = __hip_atomic_load(src + o * kWordsPerWg + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                else v = src[o * kWordsPerWg + i];
                bad += v != ((unsigned)(p - 1) << 20 | o << 10 | (unsigned)i);
            }
            const unsigned w = (unsigned)p << 20 | b << 10 | (unsigned)i;
            if (MODE == 1) __hip_atomic_store(dst + b * kWordsPerWg + i, w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            else dst[b * kWordsPerWg + i] = w;
        }
        if (MODE == 1) __builtin_amdgcn_s_waitcnt(0);  // (stores issued; vmcnt(0) covers their completion at the agent-coherent level)
        grid_barrier<MODE>(ctr, (unsigned)(p + 1) * n, errors);
    }
    if (bad) atomicAdd(errors, bad);
}

int main()
{
    hipStream_t st;
    CK(hipStreamCreate(&st));
    unsigned *bufA, *bufB, *ctr, *errors;
    CK(hipMalloc(&bufA, 256 * kWordsPerWg * 4));
    CK(hipMalloc(&bufB, 256 * kWordsPerWg * 4));
    CK(hipMalloc(&ctr, 4));
    CK(hipMalloc(&errors, 4));
    CK(hipMemset(errors, 0, 4));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    const int N = 40, reps = 50;
    auto read_err = [&]() { unsigned h; CK(hipMemcpy(&h, errors, 4, hipMemcpyDeviceToHost)); CK(hipMemset(errors, 0, 4)); return h; };
    // ---- (1) graph chains ----
    for (int kind = 0; kind < 4; ++kind) {
        const int wgs = kind == 0 ? 1 : kind == 1 ? 128 : 256, threads = kind == 3 ? 1024 : 256;
        hipGraph_t g;
        hipGraphExec_t ge;
        CK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
        for (int p = 0; p < N; ++p) {
            if (kind == 0) hipLaunchKernelGGL(k_null, dim3(1), dim3(64), 0, st, (int *)bufA);
            else hipLaunchKernelGGL(k_phase, dim3(wgs), dim3(threads), 0, st, (p & 1) ? bufA : bufB, (p & 1) ? bufB : bufA, (unsigned)p, errors);
        }
        CK(hipStreamEndCapture(st, &g));
        CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        for (int i = 0; i < 5; ++i) CK(hipGraphLaunch(ge, st));
        CK(hipStreamSynchronize(st));
        read_err();
        CK(hipEventRecord(e0, st));
        for (int i = 0; i < reps; ++i) CK(hipGraphLaunch(ge, st));
        CK(hipEventRecord(e1, st));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        printf("graph chain : %-44s %6.2f us per node   (errors %u)\n",
               kind == 0 ? "null kernel, 1 workgroup" : kind == 1 ? "hand-over kernel, 128 wg x 256" : kind == 2 ? "hand-over kernel, 256 wg x 256" : "hand-over kernel, 256 wg x 1024",
               ms * 1e3 / reps / N, read_err());
        // the same chain launched eagerly
        CK(hipEventRecord(e0, st));
        for (int i = 0; i < reps; ++i)
            for (int p = 0; p < N; ++p) {
                if (kind == 0) hipLaunchKernelGGL(k_null, dim3(1), dim3(64), 0, st, (int *)bufA);
                else hipLaunchKernelGGL(k_phase, dim3(wgs), dim3(threads), 0, st, (p & 1) ? bufA : bufB, (p & 1) ? bufB : bufA, (unsigned)p, errors);
            }
        CK(hipEventRecord(e1, st));
        CK(hipEventSynchronize(e1));
        CK(hipEventElapsedTime(&ms, e0, e1));
        printf("eager chain : %-44s %6.2f us per launch (errors %u)\n", "same", ms * 1e3 / reps / N, read_err());
        CK(hipGraphExecDestroy(ge));
        CK(hipGraphDestroy(g));
    }
    // ---- (2) one persistent kernel with grid barriers ----
    for (int mode = 0; mode < 3; ++mode)
        for (int cfg = 0; cfg < 4; ++cfg) {
            const int wgs = cfg == 0 ? 128 : 256, threads = cfg <= 1 ? 256 : cfg == 2 ? 1024 : 512;
            float best = 1e9f;
            unsigned err = 0;
            for (int rep = 0; rep < 6; ++rep) {
                CK(hipMemsetAsync(ctr, 0, 4, st));
                CK(hipEventRecord(e0, st));
                if (mode == 0) hipLaunchKernelGGL(k_persistent<0>, dim3(wgs), dim3(threads), 0, st, bufA, bufB, N * 10, ctr, errors);
                else if (mode == 1) hipLaunchKernelGGL(k_persistent<1>, dim3(wgs), dim3(threads), 0, st, bufA, bufB, N * 10, ctr, errors);
                else hipLaunchKernelGGL(k_persistent<2>, dim3(wgs), dim3(threads), 0, st, bufA, bufB, N * 10, ctr, errors);
                CK(hipEventRecord(e1, st));
                CK(hipEventSynchronize(e1));
                float ms;
                CK(hipEventElapsedTime(&ms, e0, e1));
                if (rep > 0 && ms < best) best = ms;
                err += read_err();
            }
            printf("persistent  : %-20s %3d wg x %4d   %6.2f us per phase + barrier   (errors %u)\n",
                   mode == 0 ? "release/acquire" : mode == 1 ? "relaxed + sc1 data" : "no barrier (work)", wgs, threads, best * 1e3 / (N * 10), err);
        }
    return 0;
}
