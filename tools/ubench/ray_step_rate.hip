// Micro-benchmark: what one step of the packed Bresenham walk of k_ray_list (csrc/voxel.hip::walk_packed) costs on a CU, as a
// function of the workgroups resident per CU, with and without its LDS atomic, with scattered / identical addresses, and as the
// compiled loop vs a hand-scheduled 4-step block (v_cmpx exec masking, one branch per 4 steps).
//   hipcc --offload-arch=gfx950 -O3 -o ray_step_rate tools/ubench/ray_step_rate.hip && ./ray_step_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

typedef short v2s_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ v2s_t pack2(int lo, int hi) { return __builtin_bit_cast(v2s_t, (lo & 0xffff) | (hi << 16)); }

// MODE 0: no LDS atomic; 1: atomic, every lane its own direction; 2: atomic, all lanes of the workgroup the same ray;
// 3: as 1 with the hand-scheduled loop; 4: as 2 with the hand-scheduled loop; 5: as 0 with the hand-scheduled loop (atomic removed)
template <int MODE>
__global__ __launch_bounds__(256) void k_walk(uint32_t *__restrict__ out, int reps)
{
    extern __shared__ __attribute__((aligned(16))) uint32_t s_path[];
    const int tid = threadIdx.x;
    for (int i = tid; i < 8192; i += 256) s_path[i] = 0u;
    __syncthreads();
    const int g = 64, gg = 64 * 64;
    const int sx = 32, sy = 32, sz = 32, l_src = sx * gg + sy * g + sz;
    // target on the shell of radius 30 around the source: direction from a hash of the lane (or one direction for all)
    unsigned h = (MODE == 2 || MODE == 4) ? 12345u : (unsigned)(tid * 2654435761u + blockIdx.x * 40503u);
    const int face = h % 6, u = (int)((h >> 8) % 61) - 30, v = (int)((h >> 16) % 61) - 30;
    int tx = sx, ty = sy, tz = sz;
    if (face == 0) { tx += 30; ty += u; tz += v; } else if (face == 1) { tx -= 30; ty += u; tz += v; }
    else if (face == 2) { ty += 30; tx += u; tz += v; } else if (face == 3) { ty -= 30; tx += u; tz += v; }
    else if (face == 4) { tz += 30; tx += u; ty += v; } else { tz -= 30; tx += u; ty += v; }
    const int d0 = abs(tx - sx), d1 = abs(ty - sy), d2 = abs(tz - sz);
    const int da = max(max(d0, d1), d2);
    const bool ax = da == d0, ay = !ax && da == d1;
    const int l0 = sx < tx ? gg : -gg, l1 = sy < ty ? g : -g, l2 = sz < tz ? 1 : -1;
    const int la = ax ? l0 : (ay ? l1 : l2), lb = ax ? l1 : l0, lc = (ax || ay) ? l2 : l1;
    const int db = ax ? d1 : d0, dc = (ax || ay) ? d2 : d1;
    const v2s_t neg2da = pack2(-2 * da, -2 * da), dl = pack2(2 * db - 2 * da, 2 * dc - 2 * da), lbc = pack2(lb, lc);
    const int kp = la + lb + lc - (1 << 24);
    uint32_t acc = 0;
    for (int r = 0; r < reps; ++r) {
        int W = l_src + (da << 24);
        v2s_t P = pack2(2 * db - da, 2 * dc - da);
        if (MODE <= 2) {
            while (W >= (1 << 24)) {
                const v2s_t m = P >> (v2s_t)(15);
                P = m * neg2da + (P + dl);
                W = __builtin_amdgcn_sdot2(lbc, m, W, false) + kp;
                if (MODE == 0) acc ^= (uint32_t)W;
                else atomicOr(&s_path[(W >> 5) & 0x7ffff], 1u << (W & 31));
            }
        } else {
            // four steps per trip; a lane that has taken its last step drops out of EXEC (v_cmpx), the branch is taken while any lane is
            // left.  pk_add sits between pk_ashr and its consumers, the dot product between pk_add and pk_mad (packed-result forwarding)
            int Pi = __builtin_bit_cast(int, P), m_, a_, b_;
            const int n2 = __builtin_bit_cast(int, neg2da), dli = __builtin_bit_cast(int, dl), lbi = __builtin_bit_cast(int, lbc);
            const int lim = 1 << 24;
            uint64_t saved;
#define STEP_ASM(ATOM)                                                  \
            "v_pk_ashrrev_i16 %[m], 15, %[P] op_sel_hi:[0,1]\n"         \
            "v_pk_add_u16 %[P], %[P], %[dl]\n"                          \
            "v_dot2c_i32_i16 %[W], %[lbc], %[m]\n"                      \
            "v_pk_mad_u16 %[P], %[m], %[n2], %[P]\n"                    \
            "v_add_u32 %[W], %[W], %[kp]\n"                             \
            "v_lshrrev_b32 %[a], 3, %[W]\n"                             \
            "v_lshlrev_b32 %[b], %[W], 1\n"                             \
            "v_and_b32 %[a], 0x1ffffc, %[a]\n"                          \
            ATOM                                                        \
            "v_cmpx_le_i32 vcc, %[lim], %[W]\n"
#define WALK_ASM(ATOM)                                                  \
            asm volatile(                                               \
                "s_mov_b64 %[sv], exec\n"                               \
                "v_cmpx_le_i32 vcc, %[lim], %[W]\n"                     \
                "s_cbranch_execz 1f\n"                                  \
                "0:\n"                                                  \
                STEP_ASM(ATOM) STEP_ASM(ATOM) STEP_ASM(ATOM) STEP_ASM(ATOM) \
                "s_cbranch_execnz 0b\n"                                 \
                "1:\n"                                                  \
                "s_mov_b64 exec, %[sv]\n"                               \
                : [W] "+v"(W), [P] "+v"(Pi), [m] "=&v"(m_), [a] "=&v"(a_), [b] "=&v"(b_), [sv] "=&s"(saved)                  \
                : [dl] "v"(dli), [lbc] "v"(lbi), [n2] "v"(n2), [kp] "v"(kp), [lim] "s"(lim)                                 \
                : "vcc", "memory")
            if (MODE == 5) { WALK_ASM("v_xor_b32 %[b], %[a], %[b]\n"); acc ^= (uint32_t)b_; }
            else WALK_ASM("ds_or_b32 %[a], %[b]\n");
        }
    }
    __syncthreads();
    out[blockIdx.x * 256 + tid] = acc ^ s_path[tid] ^ s_path[tid + 256];
}

template <int MODE> void run(const char *name, uint32_t *out)
{
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    (void)hipFuncSetAttribute((const void *)k_walk<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, 32768);
    printf("%-58s", name);
    for (int per_cu = 1; per_cu <= 4; ++per_cu) {
        const int blocks = 256 * per_cu, reps = 64;
        k_walk<MODE><<<blocks, 256, 32768>>>(out, 2);
        (void)hipDeviceSynchronize();
        (void)hipEventRecord(e0); k_walk<MODE><<<blocks, 256, 32768>>>(out, reps); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        const double steps = 30.0 * reps;  // per wave
        const double cyc = ms * 1e-3 * 2.1e9 / steps;  // per step of every resident wave (one "round")
        printf("  %d WG/CU: %6.1f cyc/round = %5.1f per wave-step per SIMD", per_cu, cyc, cyc / per_cu);
    }
    printf("\n");
}

int main()
{
    uint32_t *out; (void)hipMalloc(&out, 1024 * 256 * 4);
    run<0>("compiled loop, no atomic", out);
    run<1>("compiled loop, ds_or scattered", out);
    run<2>("compiled loop, ds_or one address per workgroup", out);
    run<5>("4-step asm block, no atomic", out);
    run<3>("4-step asm block, ds_or scattered", out);
    run<4>("4-step asm block, ds_or one address per workgroup", out);
    uint32_t hsum = 0, *h = (uint32_t *)malloc(1024 * 256 * 4);
    (void)hipMemcpy(h, out, 1024 * 256 * 4, hipMemcpyDeviceToHost);
    for (int i = 0; i < 1024 * 256; ++i) hsum ^= h[i];
    printf("(checksum %08x)\n", hsum);
    return 0;
}
