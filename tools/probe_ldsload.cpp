// probe_ldsload.cpp - where does `buffer_load_dwordx4 ... lds` put each lane's 16 bytes?  (gfx950)
// build: hipcc --offload-arch=gfx950 -O2 probe_ldsload.cpp -o probe_ldsload
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void k(const uint32_t* src, uint32_t* out, int nbytes) {
    extern __shared__ char smem[];
    for (int i = threadIdx.x; i < 2048; i += blockDim.x) ((uint32_t*)smem)[i] = 0xdeadbeef;
    __syncthreads();
    __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, nbytes, 0x00020000);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // lane l fetches global chunk (63 - l) + 64*wave; wave w targets LDS base 2048*w + 256
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)(smem + 2048 * wave + 256), 16,
                                             ((63 - lane) + 64 * wave) * 16, 0, 0, 0);
    // second instruction with an immediate offset of 1024
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)(smem + 2048 * wave + 256), 16,
                                             (lane + 64 * wave) * 16, 4096, 1024, 0);
    __builtin_amdgcn_s_waitcnt(0);
    __syncthreads();
    for (int i = threadIdx.x; i < 2048; i += blockDim.x) out[i] = ((uint32_t*)smem)[i];
}
int main() {
    std::vector<uint32_t> h(8192);
    for (int i = 0; i < 8192; ++i) h[i] = i;   // dword index
    uint32_t *d, *o;
    hipMalloc(&d, 8192 * 4);
    hipMalloc(&o, 2048 * 4);
    hipMemcpy(d, h.data(), 8192 * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(128), 8192, 0, d, o, 8192 * 4);
    std::vector<uint32_t> r(2048);
    hipMemcpy(r.data(), o, 2048 * 4, hipMemcpyDeviceToHost);
    printf("err=%s\n", hipGetErrorString(hipGetLastError()));
    for (int w = 0; w < 2; ++w) {
        printf("wave %d region (dwords, every 4th = one 16-B slot's first dword):\n", w);
        for (int s = 0; s < 128; ++s) {
            uint32_t v = r[w * 512 + s * 4];
            if (v == 0xdeadbeef) printf(" ----");
            else printf(" %4u", v / 4);   // source chunk index
            if (s % 16 == 15) printf("\n");
        }
        // intra-slot order check for slot 16
        printf(" slot16 dwords: %u %u %u %u\n", r[w * 512 + 64], r[w * 512 + 65], r[w * 512 + 66], r[w * 512 + 67]);
    }
    return 0;
}
