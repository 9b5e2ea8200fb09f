// gfx950 (MI355X): which packed VOP3P operand forms give wrong results while ANOTHER kernel's waves issue matrix instructions on the same SIMD?
//
// Round 6: level0_prep's tap-window build wrote wrong pixels whenever a matrix-instruction kernel of another stream shared its CUs.  The wrong value
// was traced (assembly-level probes: profiles/r06_prep_concurrency.txt) to ONE instruction, `v_pk_add_f32 v[36:37], v[36:37], v[4:5] op_sel:[0,1]`:
// in lanes 48-63 its LOW half came out as src0 + 0 — the operand selected by op_sel (the HIGH register of src1) read as zero; the high half, which reads
// the same register, was right.  This program runs such instructions in a loop, alone and beside a kernel that does nothing but matrix instructions,
// and compares every result with the same arithmetic done by plain single-register instructions.
//
//   hipcc --offload-arch=gfx950 -O2 -o pk_opsel_probe pk_opsel_probe.hip && ./pk_opsel_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef float f2 __attribute__((ext_vector_type(2)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f4 __attribute__((ext_vector_type(4)));

// KIND 1: two 16x16x32 matrix instructions per iteration, 2: two vector FMAs per iteration
template <int KIND>
__global__ __launch_bounds__(256) void aggressor(float* out, int iters) {
    h8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(threadIdx.x * 0.001f + i); b[i] = (_Float16)(1.0f + i * 0.01f); }
    f4 c0 = {0, 0, 0, 0}, c1 = c0;
    for (int i = 0; i < iters; ++i) {
        if (KIND == 1) {
            c0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c1, 0, 0, 0);
        } else {
            asm volatile("v_fma_f32 %0, %0, %0, %0\n v_fma_f32 %1, %1, %1, %1" : "+v"(c0[0]), "+v"(c1[0]));
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = c0[0] + c1[1];
}

// plain single-register arithmetic the compiler cannot pack
__device__ __forceinline__ float sadd(float x, float y) { float r; asm volatile("v_add_f32 %0, %1, %2" : "=v"(r) : "v"(x), "v"(y)); return r; }
__device__ __forceinline__ float smul(float x, float y) { float r; asm volatile("v_mul_f32 %0, %1, %2" : "=v"(r) : "v"(x), "v"(y)); return r; }
__device__ __forceinline__ float sfma(float x, float y, float z) { float r; asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(r) : "v"(x), "v"(y), "v"(z)); return r; }

#define NCASES 25
template <int ID>
__device__ __forceinline__ void run_case(f2 a, f2 b, f2 c, f2 sg, f2& r, f2& e) {
    f2 s = {__int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(sg.x))), __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(sg.y)))};                  // wave-uniform pair for the SGPR-source cases
    if constexpr (ID == 0)  { asm volatile("v_pk_add_f32 %0, %1, %2 op_sel:[0,1]" : "=v"(r) : "v"(a), "v"(b));                  e = f2{sadd(a.x, b.y), sadd(a.y, b.y)}; }
    if constexpr (ID == 1)  { asm volatile("v_pk_add_f32 %0, %1, %2 op_sel:[1,0]" : "=v"(r) : "v"(a), "v"(b));                  e = f2{sadd(a.y, b.x), sadd(a.y, b.y)}; }
    if constexpr (ID == 2)  { asm volatile("v_pk_add_f32 %0, %1, %2 op_sel_hi:[1,0]" : "=v"(r) : "v"(a), "v"(b));               e = f2{sadd(a.x, b.x), sadd(a.y, b.x)}; }
    if constexpr (ID == 3)  { asm volatile("v_pk_add_f32 %0, %1, %2 op_sel_hi:[0,1]" : "=v"(r) : "v"(a), "v"(b));               e = f2{sadd(a.x, b.x), sadd(a.x, b.y)}; }
    if constexpr (ID == 4)  { asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel:[0,1]" : "=v"(r) : "v"(a), "v"(b));                  e = f2{smul(a.x, b.y), smul(a.y, b.y)}; }
    if constexpr (ID == 5)  { asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,1,0]" : "=v"(r) : "v"(a), "v"(b), "v"(c));    e = f2{sfma(a.x, b.y, c.x), sfma(a.y, b.y, c.y)}; }
    if constexpr (ID == 6)  { asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,0,1]" : "=v"(r) : "v"(a), "v"(b), "v"(c));    e = f2{sfma(a.x, b.x, c.y), sfma(a.y, b.y, c.y)}; }
    if constexpr (ID == 7)  { asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel_hi:[1,1,0]" : "=v"(r) : "v"(a), "v"(b), "v"(c)); e = f2{sfma(a.x, b.x, c.x), sfma(a.y, b.y, c.x)}; }
    if constexpr (ID == 8)  { asm volatile("v_pk_add_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));                               e = f2{sadd(a.x, b.x), sadd(a.y, b.y)}; }
    if constexpr (ID == 9)  { asm volatile("v_pk_add_f32 %0, %1, %2 op_sel:[1,1]" : "=v"(r) : "v"(a), "v"(b));                  e = f2{sadd(a.y, b.y), sadd(a.y, b.y)}; }
    if constexpr (ID == 10) { asm volatile("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0]" : "=v"(r) : "v"(a), "v"(b));  e = f2{sadd(a.x, b.y), sadd(a.y, b.x)}; }
    if constexpr (ID == 11) { asm volatile("v_pk_mov_b32 %0, %1, %2 op_sel:[1,0]" : "=v"(r) : "v"(a), "v"(b));                  e = f2{a.y, b.x}; }
    if constexpr (ID == 12) { asm volatile("v_pk_add_f32 %0, %1, %2 op_sel:[0,1]" : "=v"(r) : "v"(a), "s"(s));                  e = f2{sadd(a.x, s.y), sadd(a.y, s.y)}; }
    if constexpr (ID == 13) { asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel:[1,0]" : "=v"(r) : "s"(s), "v"(b));                  e = f2{smul(s.y, b.x), smul(s.y, b.y)}; }
    if constexpr (ID == 14) { asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,1,0]" : "=v"(r) : "s"(s), "v"(b), "v"(c));    e = f2{sfma(s.x, b.y, c.x), sfma(s.y, b.y, c.y)}; }   // the form in dec3_synth_kernel
    if constexpr (ID == 15) { asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0]" : "=v"(r) : "s"(s), "v"(b));  e = f2{smul(s.x, b.y), smul(s.y, b.x)}; }             // the form in level0_prep_kernel<*, false>
    if constexpr (ID == 16) { asm volatile("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(r) : "v"(a), "v"(b));     e = f2{sadd(a.x, -b.x), sadd(a.y, -b.y)}; }
    if constexpr (ID == 17) { asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[0,1]" : "=v"(r) : "v"(a), "v"(b));               e = f2{smul(a.x, b.x), smul(a.x, b.y)}; }
    if constexpr (ID == 18) { asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel_hi:[1,0,1]" : "=v"(r) : "v"(a), "s"(s), "v"(c)); e = f2{sfma(a.x, s.x, c.x), sfma(a.y, s.x, c.y)}; }   // very common in the product
    if constexpr (ID == 20) { asm volatile("v_pk_mul_f32 %0, 2.0, %1 op_sel:[0,1]" : "=v"(r) : "v"(b));                           e = f2{smul(2.0f, b.y), smul(2.0f, b.y)}; }
    if constexpr (ID == 21) { asm volatile("v_pk_fma_f32 %0, %1, 1.0, %2 op_sel:[0,0,1] op_sel_hi:[1,0,1]" : "=v"(r) : "v"(a), "v"(c)); e = f2{sfma(a.x, 1.0f, c.y), sfma(a.y, 1.0f, c.y)}; }   // the fma form that failed in the kernel
    if constexpr (ID == 22) { asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,0,1]" : "=v"(r) : "s"(s), "v"(b), "v"(c));    e = f2{sfma(s.x, b.x, c.y), sfma(s.y, b.y, c.y)}; }
    if constexpr (ID == 23) { asm volatile("v_pk_mov_b32 %0, %1, %2 op_sel:[0,1]" : "=v"(r) : "v"(a), "v"(b));                  e = f2{a.x, b.y}; }
    if constexpr (ID == 24) { asm volatile("v_pk_add_f32 %0, %1, %2 op_sel:[0,1]" : "=v"(r) : "v"(a), "v"(b));                  e = f2{sadd(a.x, b.y), sadd(a.y, b.y)}; }   // case 0 again, last
    if constexpr (ID == 19) { asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,0,0]" : "=v"(r) : "v"(a), "v"(b), "v"(c));    e = f2{sfma(a.y, b.x, c.x), sfma(a.y, b.y, c.y)}; }
}
static const char* case_text[NCASES] = {
    "v_pk_add_f32 D, A, B op_sel:[0,1]", "v_pk_add_f32 D, A, B op_sel:[1,0]", "v_pk_add_f32 D, A, B op_sel_hi:[1,0]", "v_pk_add_f32 D, A, B op_sel_hi:[0,1]",
    "v_pk_mul_f32 D, A, B op_sel:[0,1]", "v_pk_fma_f32 D, A, B, C op_sel:[0,1,0]", "v_pk_fma_f32 D, A, B, C op_sel:[0,0,1]", "v_pk_fma_f32 D, A, B, C op_sel_hi:[1,1,0]",
    "v_pk_add_f32 D, A, B", "v_pk_add_f32 D, A, B op_sel:[1,1]", "v_pk_add_f32 D, A, B op_sel:[0,1] op_sel_hi:[1,0]", "v_pk_mov_b32 D, A, B op_sel:[1,0]",
    "v_pk_add_f32 D, A, S op_sel:[0,1]", "v_pk_mul_f32 D, S, B op_sel:[1,0]", "v_pk_fma_f32 D, S, B, C op_sel:[0,1,0]", "v_pk_mul_f32 D, S, B op_sel:[0,1] op_sel_hi:[1,0]",
    "v_pk_add_f32 D, A, B neg_lo:[0,1] neg_hi:[0,1]", "v_pk_mul_f32 D, A, B op_sel_hi:[0,1]", "v_pk_fma_f32 D, A, S, C op_sel_hi:[1,0,1]", "v_pk_fma_f32 D, A, B, C op_sel:[1,0,0]",
    "v_pk_mul_f32 D, 2.0, B op_sel:[0,1]", "v_pk_fma_f32 D, A, 1.0, C op_sel:[0,0,1] op_sel_hi:[1,0,1]", "v_pk_fma_f32 D, S, B, C op_sel:[0,0,1]", "v_pk_mov_b32 D, A, B op_sel:[0,1]",
    "v_pk_add_f32 D, A, B op_sel:[0,1]   (again, last)"};

// err[id * 8 + 0..3]: low-half mismatches per lane quarter, + 4..7: high-half mismatches per lane quarter
template <int ID>
__global__ __launch_bounds__(256) void victim(uint32_t* err, int iters) {
    const float fpx = (float)(blockIdx.x * 64 + (threadIdx.x & 63)), fpy = (float)(blockIdx.y * 4 + (threadIdx.x >> 6));
    f2 b = {fpx + 0.5f, fpy + 1.0f}, sg = {0.75f + (float)blockIdx.y, 1.25f + (float)blockIdx.x};
    asm volatile("" : "+v"(b));
    uint32_t e_lo = 0, e_hi = 0;
    for (int i = 0; i < iters; ++i) {
        f2 a = {(float)i * 0.5f - fpx * 0.001f, (float)i * 0.25f + fpy * 0.003f}, c = {(float)i - 7.0f, fpx * 0.125f + 3.0f};
        asm volatile("" : "+v"(a), "+v"(c));
        f2 r = {0.0f, 0.0f}, e = {0.0f, 0.0f};
        run_case<ID>(a, b, c, sg, r, e);
        e_lo += (__float_as_uint(r.x) != __float_as_uint(e.x));
        e_hi += (__float_as_uint(r.y) != __float_as_uint(e.y));
    }
    const int q = (threadIdx.x & 63) / 16;
    if (e_lo) atomicAdd(&err[ID * 8 + q], e_lo);
    if (e_hi) atomicAdd(&err[ID * 8 + 4 + q], e_hi);
}

// every case gets its own aggressor launch (started first, outlasting the case's four victim launches) and a device synchronisation behind it
template <int ID>
static void launch_all(uint32_t* derr, float* dout, int beside, hipStream_t sv, hipStream_t sa) {
    if (beside == 1) hipLaunchKernelGGL(aggressor<1>, dim3(512), dim3(256), 0, sa, dout, 1200000);
    if (beside == 2) hipLaunchKernelGGL(aggressor<2>, dim3(512), dim3(256), 0, sa, dout, 2400000);
    for (int k = 0; k < 4; ++k) hipLaunchKernelGGL((victim<ID>), dim3(60, 540), dim3(256), 0, sv, derr, 256);
    (void)hipDeviceSynchronize();
    if constexpr (ID + 1 < NCASES) launch_all<ID + 1>(derr, dout, beside, sv, sa);
}

int main() {
    uint32_t* derr; float* dout;
    if (hipMalloc(&derr, NCASES * 32) != hipSuccess || hipMalloc(&dout, 512 * 256 * 4) != hipSuccess) return 2;
    hipStream_t sv, sa;
    (void)hipStreamCreate(&sv); (void)hipStreamCreate(&sa);
    static uint32_t h[3][NCASES * 8];
    const char* env[3] = {"alone", "beside matrix instr.", "beside vector FMAs"};
    for (int beside = 0; beside < 3; ++beside) {
        (void)hipMemset(derr, 0, NCASES * 32);
        (void)hipDeviceSynchronize();
        launch_all<0>(derr, dout, beside, sv, sa);
        (void)hipDeviceSynchronize();
        (void)hipMemcpy(h[beside], derr, NCASES * 32, hipMemcpyDeviceToHost);
    }
    const double ops = 4.0 * 60 * 540 * 256 * 256;
    printf("%.2e lane-results per case and run; mismatches with single-register arithmetic as low half [lanes 0-15, 16-31, 32-47, 48-63] / high half [...]\n", ops);
    int affected = 0;
    for (int id = 0; id < NCASES; ++id) {
        for (int beside = 0; beside < 3; ++beside) {
            const uint32_t* e = h[beside] + id * 8;
            printf("%-52s %-21s low [%u, %u, %u, %u]  high [%u, %u, %u, %u]\n", beside ? "" : case_text[id], env[beside], e[0], e[1], e[2], e[3], e[4], e[5], e[6], e[7]);
        }
        uint64_t s1 = 0; for (int k = 0; k < 8; ++k) s1 += h[1][id * 8 + k];
        affected += s1 != 0;
    }
    printf("%d of %d operand forms affected beside matrix instructions\n", affected, NCASES);
    return 0;
}
