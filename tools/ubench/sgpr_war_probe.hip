// Does a VALU instruction that reads an SGPR (as data, or as its lane mask) see a value an SALU instruction writes right BEHIND it — in its
// last quarter (lanes 48-63) — when waves of a matrix-instruction kernel share the SIMD?   hipcc --offload-arch=gfx950 -O2 -o sgpr_war_probe sgpr_war_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f2 __attribute__((ext_vector_type(2)));

__global__ __launch_bounds__(256) void aggressor(float* out, int iters) {
    h8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(threadIdx.x * 0.001f + i); b[i] = (_Float16)(1.0f + i * 0.01f); }
    f4 c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
    for (int i = 0; i < iters; ++i) {
        c0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c1, 0, 0, 0);
        c2 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c2, 0, 0, 0);
        c3 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c3, 0, 0, 0);
    }
    out[blockIdx.x * 256 + threadIdx.x] = c0[0] + c1[1] + c2[2] + c3[3];
}

// NOPS: wait states between the VALU read and the SALU overwrite
template <int NOPS>
__global__ __launch_bounds__(256) void victim(uint32_t* err, int iters) {
    uint32_t e_data = 0, e_mask = 0, e_pk = 0, e_raw = 0, e_vv = 0, e_vc = 0;
    const float one = 1.0f, two = 2.0f;
    for (int i = 0; i < iters; ++i) {
        uint32_t r;
        float m;
        f2 pk = {0.0f, 0.0f};
        asm volatile(
            "s_mov_b32 s20, 0x11111111\n\t"
            "s_nop 7\n\t"
            "v_mov_b32 %0, s20\n\t"                          // data operand
            ".rept %3\n\t s_nop 0\n\t .endr\n\t"
            "s_mov_b32 s20, 0x22222222\n\t"
            "s_nop 7\n\t"
            "s_mov_b64 s[22:23], -1\n\t"
            "s_nop 7\n\t"
            "v_cndmask_b32 %1, %4, %5, s[22:23]\n\t"        // lane mask
            ".rept %3\n\t s_nop 0\n\t .endr\n\t"
            "s_mov_b64 s[22:23], 0\n\t"
            "s_nop 7\n\t"
            "s_mov_b32 s24, 1.0\n\t"
            "s_mov_b32 s25, 1.0\n\t"
            "s_nop 7\n\t"
            "v_pk_add_f32 %2, %2, s[24:25]\n\t"             // packed data operand
            ".rept %3\n\t s_nop 0\n\t .endr\n\t"
            "s_mov_b32 s24, 4.0\n\t"
            "s_mov_b32 s25, 4.0\n\t"
            "s_nop 7\n\t"
            : "=v"(r), "=v"(m), "+v"(pk)
            : "n"(NOPS), "v"(one), "v"(two)
            : "s20", "s22", "s23", "s24", "s25", "vcc");
        // read-after-write: a VALU compare writes a scalar lane mask, an SALU instruction consumes it at once
        float raw;
        uint32_t big = 1000u + threadIdx.x;
        asm volatile(
            "s_mov_b64 s[26:27], 0\n\t"
            "s_mov_b64 s[28:29], -1\n\t"
            "s_nop 7\n\t"
            "v_cmp_gt_u32_e64 s[26:27], %3, 5\n\t"        // true in every lane: s[26:27] <- -1
            ".rept %4\n\t s_nop 0\n\t .endr\n\t"
            "s_and_b64 s[28:29], s[26:27], s[28:29]\n\t"   // SALU reads the pair the compare is writing
            "s_nop 7\n\t"
            "v_cndmask_b32 %0, %1, %2, s[28:29]\n\t"
            "s_nop 7\n\t"
            : "=v"(raw) : "v"(one), "v"(two), "v"(big), "n"(NOPS) : "s26", "s27", "s28", "s29");
        e_raw += (raw != two);
        // VALU writes a scalar lane mask, the NEXT VALU instruction reads it as its mask (the gfx940+ co-execution hazard: 2 wait states)
        float vv;
        asm volatile(
            "s_mov_b64 s[30:31], 0\n\t"
            "s_nop 7\n\t"
            "v_cmp_gt_u32_e64 s[30:31], %3, 5\n\t"
            ".rept %4\n\t s_nop 0\n\t .endr\n\t"
            "v_cndmask_b32 %0, %1, %2, s[30:31]\n\t"
            "s_nop 7\n\t"
            : "=v"(vv) : "v"(one), "v"(two), "v"(big), "n"(NOPS) : "s30", "s31");
        e_vv += (vv != two);
        // ... and through VCC (e32 compare -> e32 select)
        float vc;
        asm volatile(
            "s_mov_b64 vcc, 0\n\t"
            "s_nop 7\n\t"
            "v_cmp_gt_u32_e32 vcc, 5, %3\n\t"             // 5 > big: false in every lane ... use lt instead below
            "s_nop 7\n\t"
            "s_mov_b64 vcc, 0\n\t"
            "s_nop 7\n\t"
            "v_cmp_lt_u32_e32 vcc, 5, %3\n\t"             // 5 < big: true in every lane
            ".rept %4\n\t s_nop 0\n\t .endr\n\t"
            "v_cndmask_b32_e32 %0, %1, %2, vcc\n\t"
            "s_nop 7\n\t"
            : "=v"(vc) : "v"(one), "v"(two), "v"(big), "n"(NOPS) : "vcc");
        e_vc += (vc != two);
        e_data += (r != 0x11111111u);
        e_mask += (m != two);
        e_pk += (pk[0] != 1.0f) + (pk[1] != 1.0f);
    }
    const int t = blockIdx.x * 256 + threadIdx.x;
    err[t * 6] = e_data; err[t * 6 + 1] = e_mask; err[t * 6 + 2] = e_pk; err[t * 6 + 3] = e_raw; err[t * 6 + 4] = e_vv; err[t * 6 + 5] = e_vc;
}

int main() {
    const int WG = 256 * 7, AG = 256 * 2;
    uint32_t* err; float* aout;
    hipMalloc(&err, sizeof(uint32_t) * 6 * WG * 256);
    hipMalloc(&aout, sizeof(float) * AG * 256);
    hipStream_t s1, s2;
    hipStreamCreateWithFlags(&s1, hipStreamNonBlocking); hipStreamCreateWithFlags(&s2, hipStreamNonBlocking);
    std::vector<uint32_t> h(6 * WG * 256);
    auto report = [&](const char* what) {
        hipMemcpy(h.data(), err, h.size() * 4, hipMemcpyDeviceToHost);
        unsigned long long q[6][4] = {{0}};
        for (int t = 0; t < WG * 256; ++t) for (int k = 0; k < 6; ++k) q[k][(t & 63) >> 4] += h[t * 6 + k];
        printf("%-44s errors by lane quarter — data operand: %llu %llu %llu %llu | lane mask: %llu %llu %llu %llu | packed operand: %llu %llu %llu %llu | compare -> SALU read: %llu %llu %llu %llu | compare -> select (SGPR pair): %llu %llu %llu %llu | (VCC): %llu %llu %llu %llu\n", what,
               q[0][0], q[0][1], q[0][2], q[0][3], q[1][0], q[1][1], q[1][2], q[1][3], q[2][0], q[2][1], q[2][2], q[2][3], q[3][0], q[3][1], q[3][2], q[3][3], q[4][0], q[4][1], q[4][2], q[4][3], q[5][0], q[5][1], q[5][2], q[5][3]);
    };
    for (int rep = 0; rep < 2; ++rep) {
        hipLaunchKernelGGL(victim<0>, dim3(WG), dim3(256), 0, s2, err, 2000); hipDeviceSynchronize(); report("victim alone, 0 wait states");
        hipLaunchKernelGGL(aggressor, dim3(AG), dim3(256), 0, s1, aout, 400000);
        hipLaunchKernelGGL(victim<0>, dim3(WG), dim3(256), 0, s2, err, 2000); hipDeviceSynchronize(); report("victim beside MFMA waves, 0 wait states");
        hipLaunchKernelGGL(aggressor, dim3(AG), dim3(256), 0, s1, aout, 400000);
        hipLaunchKernelGGL(victim<2>, dim3(WG), dim3(256), 0, s2, err, 2000); hipDeviceSynchronize(); report("victim beside MFMA waves, 2 wait states");
        hipLaunchKernelGGL(aggressor, dim3(AG), dim3(256), 0, s1, aout, 400000);
        hipLaunchKernelGGL(victim<1>, dim3(WG), dim3(256), 0, s2, err, 2000); hipDeviceSynchronize(); report("victim beside MFMA waves, 1 wait state");
        hipLaunchKernelGGL(aggressor, dim3(AG), dim3(256), 0, s1, aout, 400000);
        hipLaunchKernelGGL(victim<8>, dim3(WG), dim3(256), 0, s2, err, 2000); hipDeviceSynchronize(); report("victim beside MFMA waves, 8 wait states");
    }
    return 0;
}
