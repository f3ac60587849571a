// Issue-rate micro-benchmarks for the instructions the blind-rotation kernels are made of (development aid, gfx950).
// Every test is a loop over an unrolled block of INDEPENDENT copies of one instruction (or a small mix); the grid is
// 256 CUs x 4 SIMDs x W waves, and the figure printed is shader cycles per wave-instruction per SIMD
// (s_memtime / s_memrealtime give the shader clock).  Build: hipcc -O3 --offload-arch=gfx950 valu_rates.hip -o build/valu_rates
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <string>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

constexpr int kIters = 20000;

typedef unsigned v2u_t __attribute__((ext_vector_type(2)));
typedef double d2v __attribute__((ext_vector_type(2)));

// 8 independent register sets; BODY uses a (dst/acc), b, c
#define R8(M) M(0) M(1) M(2) M(3) M(4) M(5) M(6) M(7)

template <int OP>
__global__ __launch_bounds__(256) void k_rate(double* out, unsigned long long* cyc, int iters) {
    __shared__ __align__(16) double2 lds[4 * 640];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    double a[8], b[8], c[8];
    typedef double d4v __attribute__((ext_vector_type(4)));
    d4v acc[4] = {};
    int ia[8], ib[8];
#pragma unroll
    for (int i = 0; i < 8; i++) {
        a[i] = 1.0 + 1e-9 * (tid + i);
        b[i] = 1.0 - 1e-12 * (tid + 3 * i);
        c[i] = 1e-13 * (i + 1);
        ia[i] = tid * 7 + i;
        ib[i] = tid + 11 * i;
    }
    double2* my = lds + wave * 640;
    for (int i = lane; i < 640; i += 64) my[i] = make_double2(1.0, 2.0);
    __syncthreads();
    const unsigned long long t0 = __builtin_readcyclecounter();
#pragma unroll 1
    for (int it = 0; it < iters; it++) {
        if constexpr (OP == 0) {  // v_fma_f64, 3 distinct vgpr sources
#define M(i) asm volatile("v_fma_f64 %0, %1, %2, %0" : "+v"(a[i]) : "v"(b[i]), "v"(c[i]));
            R8(M) R8(M)
#undef M
        } else if constexpr (OP == 1) {
#define M(i) asm volatile("v_add_f64 %0, %0, %1" : "+v"(a[i]) : "v"(c[i]));
            R8(M) R8(M)
#undef M
        } else if constexpr (OP == 2) {
#define M(i) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(a[i]) : "v"(b[i]));
            R8(M) R8(M)
#undef M
        } else if constexpr (OP == 3) {
#define M(i) asm volatile("v_cvt_f64_i32 %0, %1" : "=v"(a[i]) : "v"(ia[i]));
            R8(M) R8(M)
#undef M
        } else if constexpr (OP == 4) {
#define M(i) asm volatile("v_bfe_i32 %0, %1, 11, 7" : "=v"(ia[i]) : "v"(ib[i]));
            R8(M) R8(M)
#undef M
        } else if constexpr (OP == 5) {
#define M(i) asm volatile("v_add_u32 %0, %0, %1" : "+v"(ia[i]) : "v"(ib[i]));
            R8(M) R8(M)
#undef M
        } else if constexpr (OP == 6) {  // dpp mov row_ror:8 with bank mask (the bit-3 exchange)
#define M(i) asm volatile("v_mov_b32_dpp %0, %1 row_ror:8 row_mask:0xf bank_mask:0x3" : "+v"(ia[i]) : "v"(ib[i]));
            R8(M) R8(M)
#undef M
        } else if constexpr (OP == 7) {
#define M(i) asm volatile("v_permlane32_swap_b32 %0, %1" : "+v"(ia[i]), "+v"(ib[i]));
            R8(M) R8(M)
#undef M
        } else if constexpr (OP == 8) {
#define M(i) asm volatile("v_permlane16_swap_b32 %0, %1" : "+v"(ia[i]), "+v"(ib[i]));
            R8(M) R8(M)
#undef M
        } else if constexpr (OP == 9) {
#define M(i) asm volatile("v_max_f64 %0, %0, |%1|" : "+v"(a[i]) : "v"(b[i]));
            R8(M) R8(M)
#undef M
        } else if constexpr (OP == 10) {
#define M(i) asm volatile("v_mov_b32 %0, %1" : "=v"(ia[i]) : "v"(ib[i]));
            R8(M) R8(M)
#undef M
        } else if constexpr (OP == 11) {
#define M(i) asm volatile("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(ia[i]) : "v"(ib[i]), "s"(0x5555aaaa5555aaaaull));
            R8(M) R8(M)
#undef M
        } else if constexpr (OP == 12) {  // quad_perm dpp
#define M(i) asm volatile("v_mov_b32_dpp %0, %1 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf" : "+v"(ia[i]) : "v"(ib[i]));
            R8(M) R8(M)
#undef M
        } else if constexpr (OP == 13) {
#define M(i) asm volatile("v_cvt_i32_f64 %0, %1" : "=v"(ia[i]) : "v"(a[i]));
            R8(M) R8(M)
#undef M
        } else if constexpr (OP == 14) {  // v_fmac_f64 (VOP2 form)
#define M(i) asm volatile("v_fmac_f64 %0, %1, %2" : "+v"(a[i]) : "v"(b[i]), "v"(c[i]));
            R8(M) R8(M)
#undef M
        } else if constexpr (OP == 15) {  // v_add_f64 with an sgpr-pair/inline constant operand
#define M(i) asm volatile("v_add_f64 %0, %0, 0.5" : "+v"(a[i]));
            R8(M) R8(M)
#undef M
        } else if constexpr (OP == 16) {  // v_xad_u32
#define M(i) asm volatile("v_xad_u32 %0, %0, %1, %2" : "+v"(ia[i]) : "v"(ib[i]), "v"(ib[(i + 1) & 7]));
            R8(M) R8(M)
#undef M
        } else if constexpr (OP == 17) {  // v_lshl_add_u32
#define M(i) asm volatile("v_lshl_add_u32 %0, %0, 2, %1" : "+v"(ia[i]) : "v"(ib[i]));
            R8(M) R8(M)
#undef M
        } else if constexpr (OP == 18) {  // v_mov_b64
#define M(i) asm volatile("v_mov_b64 %0, %1" : "=v"(a[i]) : "v"(b[i]));
            R8(M) R8(M)
#undef M
        } else if constexpr (OP == 19) {  // v_pk_add_f32 as a 64-bit move-ish op (rate probe of packed fp32)
#define M(i) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(a[i]) : "v"(b[i]));
            R8(M) R8(M)
#undef M
        } else if constexpr (OP == 20) {  // ds_read_b128, conflict-free (lane-contiguous)
            d2v r[8];
#define M(i) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(r[i]) : "v"((unsigned)(size_t)(my + lane) & 0xffff), "i"(i * 1024 % 8192));
            R8(M) R8(M)
#undef M
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
            for (int i = 0; i < 8; i++) a[i] += r[i].x;
        } else if constexpr (OP == 21) {  // ds_write_b128
            d2v v; v.x = a[0]; v.y = b[0];
#define M(i) asm volatile("ds_write_b128 %0, %1 offset:%2" : : "v"((unsigned)(size_t)(my + lane) & 0xffff), "v"(v), "i"(i * 1024 % 8192) : "memory");
            R8(M) R8(M)
#undef M
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        } else if constexpr (OP == 22) {  // ds_read_b32
            int r[8];
#define M(i) asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(r[i]) : "v"((unsigned)(size_t)((int*)my + lane) & 0xffff), "i"(i * 256));
            R8(M) R8(M)
#undef M
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
            for (int i = 0; i < 8; i++) ia[i] += r[i];
        } else if constexpr (OP == 23) {  // ds_add_u32 (no return)
#define M(i) asm volatile("ds_add_u32 %0, %1 offset:%2" : : "v"((unsigned)(size_t)((int*)my + lane) & 0xffff), "v"(ia[i]), "i"(i * 256) : "memory");
            R8(M) R8(M)
#undef M
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        } else if constexpr (OP == 24) {  // ds_read_b128, 8-address broadcast pattern (the second twiddle table)
            d2v r[8];
#define M(i) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(r[i]) : "v"((unsigned)(size_t)(my + (lane & 7)) & 0xffff), "i"(i * 128));
            R8(M) R8(M)
#undef M
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
            for (int i = 0; i < 8; i++) a[i] += r[i].x;
        } else if constexpr (OP == 25) {  // ds_read_b64
            double r[8];
#define M(i) asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(r[i]) : "v"((unsigned)(size_t)((double*)my + lane) & 0xffff), "i"(i * 512));
            R8(M) R8(M)
#undef M
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
            for (int i = 0; i < 8; i++) a[i] += r[i];
        } else if constexpr (OP == 26) {  // ds_write_b64
#define M(i) asm volatile("ds_write_b64 %0, %1 offset:%2" : : "v"((unsigned)(size_t)((double*)my + lane) & 0xffff), "v"(a[i]), "i"(i * 512) : "memory");
            R8(M) R8(M)
#undef M
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        } else if constexpr (OP == 27) {  // mix: 12 fma_f64 + 4 ds_read_b128 (co-issue across waves)
            d2v r[4];
#define M(i) asm volatile("v_fma_f64 %0, %1, %2, %0" : "+v"(a[i]) : "v"(b[i]), "v"(c[i]));
#define D(i) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(r[i]) : "v"((unsigned)(size_t)(my + lane) & 0xffff), "i"(i * 1024));
            M(0) M(1) M(2) D(0) M(3) M(4) M(5) D(1) M(6) M(7) M(0) D(2) M(1) M(2) M(3) D(3)
#undef M
#undef D
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            a[4] += r[0].x + r[1].x + r[2].x + r[3].x;
        } else if constexpr (OP == 28) {  // v_fma_f64 reading sources that share a VGPR bank (regs 4 apart)
#define M(i) asm volatile("v_fma_f64 %0, %0, %0, %0" : "+v"(a[i]));
            R8(M) R8(M)
#undef M
        } else if constexpr (OP == 29) {  // v_readlane_b32
            int s;
#define M(i) asm volatile("v_readlane_b32 %0, %1, 3" : "=s"(s) : "v"(ia[i])); 
            R8(M) R8(M)
#undef M
            ia[0] += s;
        } else if constexpr (OP == 30) {  // v_and_or_b32
#define M(i) asm volatile("v_and_or_b32 %0, %0, %2, %1" : "+v"(ia[i]) : "v"(ib[i]), "s"(4095));
            R8(M) R8(M)
#undef M
        } else if constexpr (OP == 31) {  // v_mul_f64 by a literal-free constant in SGPRs
            const double kk = 0.70710678118654752440;
#define M(i) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(a[i]) : "s"(kk));
            R8(M) R8(M)
#undef M
        } else if constexpr (OP == 32) {  // dependent chain of v_fma_f64 (latency)
#define M(i) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(a[0]) : "v"(b[i]), "v"(c[i]));
            R8(M) R8(M)
#undef M
        } else if constexpr (OP == 33) {  // dependent chain of v_add_u32
#define M(i) asm volatile("v_add_u32 %0, %0, %1" : "+v"(ia[0]) : "v"(ib[i]));
            R8(M) R8(M)
#undef M
        } else if constexpr (OP == 34) {  // dpp feeding an f64 add (hazard probe): mov_dpp then add that uses it
#define M(i) asm volatile("v_mov_b32_dpp %0, %1 row_ror:8 row_mask:0xf bank_mask:0x3\n\tv_add_u32 %2, %2, %0" : "+v"(ia[i]), "+v"(ib[i]), "+v"(ia[(i + 4) & 7]));
            R8(M)
#undef M
        } else if constexpr (OP == 35) {  // v_mfma_f64_16x16x4_f64 alone: 16 instructions, four independent accumulators
#define X(j) acc[j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[j], b[j], acc[j], 0, 0, 0);
            X(0) X(1) X(2) X(3) X(0) X(1) X(2) X(3) X(0) X(1) X(2) X(3) X(0) X(1) X(2) X(3)
#undef X
        } else if constexpr (OP == 36) {  // ... each followed by four independent v_fma_f64: 16 MFMA + 64 FMA (do the two pipes overlap?)
#define X(j) acc[j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[j], b[j], acc[j], 0, 0, 0);
#define M(i) asm volatile("v_fma_f64 %0, %1, %2, %0" : "+v"(a[4 + (i & 3)]) : "v"(b[i]), "v"(c[i]));
#define G(j) X(j) M(0) M(1) M(2) M(3)
            G(0) G(1) G(2) G(3) G(0) G(1) G(2) G(3) G(0) G(1) G(2) G(3) G(0) G(1) G(2) G(3)
#undef G
#undef M
#undef X
        } else if constexpr (OP == 37) {  // the 64 v_fma_f64 of OP 36 without the MFMAs
#define M(i) asm volatile("v_fma_f64 %0, %1, %2, %0" : "+v"(a[4 + (i & 3)]) : "v"(b[i]), "v"(c[i]));
#define G(j) M(0) M(1) M(2) M(3)
            G(0) G(1) G(2) G(3) G(0) G(1) G(2) G(3) G(0) G(1) G(2) G(3) G(0) G(1) G(2) G(3)
#undef G
#undef M
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    double s = 0;
    int si = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        s += a[i] + b[i];
        si += ia[i] + ib[i];
    }
    s += acc[0].x + acc[1].y + acc[2].z + acc[3].w;
    if (s == 12345.678 || si == 0x7fffffff) out[tid] = s + si;
    if (tid == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}

struct Test { int op; const char* name; int per_iter; };

template <int OP>
static void run(const char* name, int per_iter, int waves_per_simd, double* d_out, unsigned long long* d_cyc) {
    int dev_cus = 256;
    const int blocks = dev_cus * waves_per_simd;  // 256 threads = 4 waves = one per SIMD
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    hipLaunchKernelGGL(k_rate<OP>, dim3(blocks), dim3(256), 0, 0, d_out, d_cyc, 2000);
    CHECK(hipDeviceSynchronize());
    float ms1 = 0, ms3 = 0;
    unsigned long long cyc1 = 0, cyc3 = 0;
    for (int rep = 0; rep < 2; rep++) {
        const int iters = rep ? 3 * kIters : kIters;
        CHECK(hipEventRecord(e0));
        hipLaunchKernelGGL(k_rate<OP>, dim3(blocks), dim3(256), 0, 0, d_out, d_cyc, iters);
        CHECK(hipEventRecord(e1));
        CHECK(hipDeviceSynchronize());
        CHECK(hipEventElapsedTime(rep ? &ms3 : &ms1, e0, e1));
        CHECK(hipMemcpy(rep ? &cyc3 : &cyc1, d_cyc, sizeof cyc1, hipMemcpyDeviceToHost));
    }
    // slope between the N- and 3N-iteration launches: launch overhead and clock ramp cancel
    const double insts_per_simd = 2.0 * kIters * per_iter * waves_per_simd;
    const double ns = (ms3 - ms1) * 1e6 / insts_per_simd;
    printf("%-44s waves/SIMD %d  %8.3f ms  => %6.3f ns per wave-instruction per SIMD = %5.2f cycles @2.4GHz ; memtime ticks per instruction of one wave %.2f\n", name,
           waves_per_simd, ms3, ns, ns * 2.4, (double)(cyc3 - cyc1) / (2.0 * kIters * per_iter));
    CHECK(hipEventDestroy(e0));
    CHECK(hipEventDestroy(e1));
}

int main(int argc, char** argv) {
    // waves per SIMD to measure (default 1 2); 3 and 4 need the 40 KB of LDS per block to fit 3 / 4 times into a CU (they do: 160 KB)
    std::vector<int> occ;
    for (int i = 1; i < argc; i++) occ.push_back(atoi(argv[i]));
    if (occ.empty()) occ = {1, 2};
    double* d_out;
    unsigned long long* d_cyc;
    CHECK(hipMalloc(&d_out, 4096));
    CHECK(hipMalloc(&d_cyc, 64));
    for (int w : occ) {
#define T(OP, NAME, N) run<OP>(NAME, N, w, d_out, d_cyc);
        T(0, "v_fma_f64 (3 vgpr srcs)", 16)
        T(14, "v_fmac_f64", 16)
        T(28, "v_fma_f64 d,d,d,d (same reg)", 16)
        T(1, "v_add_f64", 16)
        T(15, "v_add_f64 with inline const", 16)
        T(2, "v_mul_f64", 16)
        T(31, "v_mul_f64 by sgpr pair", 16)
        T(9, "v_max_f64 with |abs|", 16)
        T(3, "v_cvt_f64_i32", 16)
        T(13, "v_cvt_i32_f64", 16)
        T(18, "v_mov_b64", 16)
        T(19, "v_pk_add_f32", 16)
        T(4, "v_bfe_i32", 16)
        T(5, "v_add_u32", 16)
        T(16, "v_xad_u32", 16)
        T(17, "v_lshl_add_u32", 16)
        T(30, "v_and_or_b32", 16)
        T(10, "v_mov_b32", 16)
        T(11, "v_cndmask_b32", 16)
        T(6, "v_mov_b32_dpp row_ror:8 bank_mask", 16)
        T(12, "v_mov_b32_dpp quad_perm", 16)
        T(34, "v_mov_b32_dpp + dependent v_add_u32 (pair)", 16)
        T(7, "v_permlane32_swap_b32", 16)
        T(8, "v_permlane16_swap_b32", 16)
        T(29, "v_readlane_b32", 16)
        T(32, "v_fma_f64 dependent chain", 16)
        T(33, "v_add_u32 dependent chain", 16)
        T(20, "ds_read_b128 lane-contiguous", 16)
        T(24, "ds_read_b128 8-address broadcast", 16)
        T(21, "ds_write_b128 lane-contiguous", 16)
        T(25, "ds_read_b64", 16)
        T(26, "ds_write_b64", 16)
        T(22, "ds_read_b32", 16)
        T(23, "ds_add_u32", 16)
        T(27, "mix 12 v_fma_f64 + 4 ds_read_b128", 16)
        T(35, "v_mfma_f64_16x16x4_f64", 16)
        T(36, "16 v_mfma_f64_16x16x4 + 64 v_fma_f64 (per 80)", 80)
        T(37, "the 64 v_fma_f64 of that mix alone", 64)
#undef T
    }
    return 0;
}
