// Where do the waves of a workgroup land?  Prints, per workgroup size, the SIMD id (HW_ID bits 5:4) of every wave of a
// few workgroups, and how many workgroups had waves w and w + 4 on the same SIMD (development aid, gfx950).
// Build: hipcc -O3 --offload-arch=gfx950 wave_simd.hip -o build/wave_simd
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__global__ void k_where(unsigned* out, int lds_bytes_used) {
    extern __shared__ unsigned char smem[];
    if (lds_bytes_used < 0) smem[threadIdx.x] = 1;  // keep the allocation
    const unsigned hw = __builtin_amdgcn_s_getreg((4 /*HW_REG_HW_ID*/) | (0 << 6) | (31 << 11));
    const int wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    if ((threadIdx.x & 63) == 0) out[(size_t)blockIdx.x * nw + wave] = hw;
    // stay resident for a moment so that the workgroups of one launch coexist as they would in a real kernel
    const unsigned long long t0 = __builtin_readcyclecounter();
    while (__builtin_readcyclecounter() - t0 < 200000) __builtin_amdgcn_s_sleep(8);
}

int main() {
    const int sizes[] = {128, 256, 512, 1024};
    for (int threads : sizes) {
        for (int lds : {0, 36 * 1024, 144 * 1024}) {
            if (lds > 64 * 1024 && threads < 512) continue;
            const int nw = threads / 64, wgs = 1024;
            unsigned* d;
            CHECK(hipMalloc(&d, sizeof(unsigned) * nw * wgs));
            if (lds > 48 * 1024) CHECK(hipFuncSetAttribute((const void*)k_where, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
            hipLaunchKernelGGL(k_where, dim3(wgs), dim3(threads), lds, 0, d, 0);
            CHECK(hipDeviceSynchronize());
            std::vector<unsigned> h(nw * wgs);
            CHECK(hipMemcpy(h.data(), d, sizeof(unsigned) * nw * wgs, hipMemcpyDeviceToHost));
            int paired = 0, spread = 0;
            for (int g = 0; g < wgs; g++) {
                bool p = nw >= 8, s = true;
                for (int w = 0; w + 4 < nw; w++) p = p && (((h[g * nw + w] >> 4) & 3) == ((h[g * nw + w + 4] >> 4) & 3));
                for (int w = 0; w < nw && w < 4; w++)
                    for (int v = 0; v < w; v++) s = s && (((h[g * nw + w] >> 4) & 3) != ((h[g * nw + v] >> 4) & 3));
                paired += p, spread += s;
            }
            printf("threads %4d lds %6d: waves w / w+4 on one SIMD in %d of %d workgroups; first four waves on distinct SIMDs in %d; e.g.", threads, lds,
                   paired, wgs, spread);
            for (int g = 0; g < 3; g++) {
                printf("  [");
                for (int w = 0; w < nw; w++) printf("%u", (h[g * nw + w] >> 4) & 3);
                printf(" cu%u se%u]", (h[g * nw] >> 8) & 15, (h[g * nw] >> 13) & 7);
            }
            printf("\n");
            CHECK(hipFree(d));
        }
    }
    return 0;
}
