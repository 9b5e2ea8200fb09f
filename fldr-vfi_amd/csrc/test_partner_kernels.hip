// Test build only (-DFLDR_TEST_HOOKS): a kernel that does nothing but OCCUPY the compute units for a while — the partner of the
// concurrency tests (tests/test_gpu_parity.py, tools/pairwise_concurrency.py).  The product kernels must give the same bits whatever
// else is issuing on their SIMDs; round 6 found one that did not (level0_prep's tap-window build: a packed-fp32 add with op_sel on
// its second source read 0 in lanes 48-63 — profiles/r06_prep_concurrency.txt), and a partner of plain vector FMAs, two workgroups per
// CU, triggered it in 8 of 8 runs where the forward's own kernels needed dozens.  The product library compiles this file to nothing.
#include "common.h"
#ifdef FLDR_TEST_HOOKS
#include "fldr_hip_test_hooks.h"
typedef _Float16 partner_h8 __attribute__((ext_vector_type(8)));
typedef float partner_f4 __attribute__((ext_vector_type(4)));

// kind 0: sleeping (holds wave slots and LDS, issues almost nothing); 1: 16x16x32 matrix instructions; 2: vector FMAs; 3: scalar adds;
// 4: LDS reads
__device__ __forceinline__ void busy_partner_body(float* out, int iters, int kind, unsigned char* partner_lds) {
    partner_h8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(threadIdx.x * 0.001f + i); b[i] = (_Float16)(1.0f + i * 0.01f); }
    partner_f4 c0 = {0, 0, 0, 0}, c1 = c0;
    if (threadIdx.x == 0) partner_lds[0] = 1;
    // one tight loop per kind (a switch inside the loop thins the instruction stream out: such a partner did not disturb anything)
    if (kind == 1) {
        for (int i = 0; i < iters; ++i) {                                   // (two per iteration: the stream beside which the defect showed in 8 of 8 runs)
            c0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c1, 0, 0, 0);
        }
    } else if (kind == 2) {
        for (int i = 0; i < iters; ++i) asm volatile("v_fma_f32 %0, %0, %0, %0\n v_fma_f32 %1, %1, %1, %1" : "+v"(c0[0]), "+v"(c1[0]));
    } else if (kind == 3) {
        for (int i = 0; i < iters; ++i) { int t; asm volatile("s_add_u32 %0, 1, 2\n s_add_u32 %0, 1, 2" : "=s"(t) : : "scc"); }      // (scc: the loop's own compare lives there)
    } else if (kind == 4) {
        for (int i = 0; i < iters; ++i) c0[0] += (float)partner_lds[(threadIdx.x * 4 + i) & 1023];
    } else {
        for (int i = 0; i < iters; ++i) __builtin_amdgcn_s_sleep(8);
    }
    out[(size_t)blockIdx.x * 256 + threadIdx.x] = c0[0] + c1[1] + (float)partner_lds[0];
}

// FP: the register footprint — the kernel allocates at least up to the named vector / accumulation register (touched once, declared clobbered): which
// physical registers the waves beside it get depends on that
template <int FP>
__global__ __launch_bounds__(256) void busy_partner_kernel(float* out, int iters, int kind) {
    extern __shared__ unsigned char partner_lds[];
    if (FP == 1) { asm volatile("v_mov_b32 v35, 0" ::: "v35"); asm volatile("v_accvgpr_write_b32 a15, 0" ::: "a15"); }
    if (FP == 2) { asm volatile("v_mov_b32 v55, 0" ::: "v55"); }
    if (FP == 3) { asm volatile("v_mov_b32 v39, 0" ::: "v39"); asm volatile("v_accvgpr_write_b32 a15, 0" ::: "a15"); }
    if (FP == 4) { asm volatile("v_mov_b32 v63, 0" ::: "v63"); }
    if (FP == 5) { asm volatile("v_accvgpr_write_b32 a31, 0" ::: "a31"); }
    if (FP == 6) { asm volatile("v_mov_b32 v47, 0" ::: "v47"); }
    if (FP == 7) { asm volatile("v_mov_b32 v31, 0" ::: "v31"); asm volatile("v_accvgpr_write_b32 a15, 0" ::: "a15"); }
    if (FP == 8) { asm volatile("v_accvgpr_write_b32 a15, 0" ::: "a15"); }
    if (FP == 9) { asm volatile("v_mov_b32 v39, 0" ::: "v39"); }
    busy_partner_body(out, iters, kind, partner_lds);
}

template <int FP>
static int busy_partner_launch(float* out, int workgroups, int lds_bytes, int iters, int kind, hipStream_t s) {
    static int attr_set = 0;
    if (!attr_set) {
        const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&busy_partner_kernel<FP>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return (int)e;
        attr_set = 1;
    }
    hipLaunchKernelGGL((busy_partner_kernel<FP>), dim3(workgroups), dim3(256), lds_bytes, s, out, iters, kind);
    FLDR_LAUNCH_RET();
}
// kind: bits 0-3 what the loop issues (above), bits 4-7 the register footprint (vector + accumulation registers per lane; tools/kernel_resources.py prints them)
extern "C" int fldr_debug_busy_partner(float* out, int workgroups, int lds_bytes, int iters, int kind, fldr_stream_t stream) {
    const int what = kind & 15, fp = (kind >> 4) & 15;
    FLDR_CHECK_ARG(out && workgroups > 0 && workgroups <= 4096 && lds_bytes >= 1024 && lds_bytes <= 160 * 1024 && iters >= 0 && kind >= 0 && what <= 4 && fp <= 9);
    hipStream_t s = fldr_s(stream);
    switch (fp) {
    case 1: return busy_partner_launch<1>(out, workgroups, lds_bytes, iters, what, s);
    case 2: return busy_partner_launch<2>(out, workgroups, lds_bytes, iters, what, s);
    case 3: return busy_partner_launch<3>(out, workgroups, lds_bytes, iters, what, s);
    case 4: return busy_partner_launch<4>(out, workgroups, lds_bytes, iters, what, s);
    case 5: return busy_partner_launch<5>(out, workgroups, lds_bytes, iters, what, s);
    case 6: return busy_partner_launch<6>(out, workgroups, lds_bytes, iters, what, s);
    case 7: return busy_partner_launch<7>(out, workgroups, lds_bytes, iters, what, s);
    case 8: return busy_partner_launch<8>(out, workgroups, lds_bytes, iters, what, s);
    case 9: return busy_partner_launch<9>(out, workgroups, lds_bytes, iters, what, s);
    default: return busy_partner_launch<0>(out, workgroups, lds_bytes, iters, what, s);
    }
}
#endif
