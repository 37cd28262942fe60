// probe: device drop_mix / drop_word against the host mirror (flash-attention-softmax-n_amd/dropout.py); prints the first fields
#include <hip/hip_runtime.h>
#include <cstdio>
#include "../flash-attention-softmax-n_amd/csrc/fasn_common.h"
using namespace fasn;
__global__ void k(uint32_t* out, uint32_t seed_lo, uint32_t seed_hi, uint32_t bh) {
    const int row = blockIdx.x, key = threadIdx.x;
    const uint32_t rb = drop_row_base(seed_lo, bh, row);
    const uint32_t y = drop_mix(rb, seed_hi, key >> 4);
    out[row * 256 + key] = drop_word(y, drop_lane(key));              // the key's field in the high half (lane-dependent form)
    uint32_t w = 0;                                                    // the pair word (compile-time form): field in half key & 1
    switch ((key & 15) >> 1) {
        case 0: w = drop_pair_word<0>(y); break; case 1: w = drop_pair_word<1>(y); break; case 2: w = drop_pair_word<2>(y); break; case 3: w = drop_pair_word<3>(y); break;
        case 4: w = drop_pair_word<4>(y); break; case 5: w = drop_pair_word<5>(y); break; case 6: w = drop_pair_word<6>(y); break; default: w = drop_pair_word<7>(y); break;
    }
    out[16 * 256 + row * 256 + key] = (key & 1) ? w : (w << 16);
}
int main() {
    uint32_t* d;
    hipMalloc(&d, 2 * 16 * 256 * 4);
    k<<<16, 256>>>(d, 12345u, 678u, 3u);
    uint32_t h[2 * 16 * 256];
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    for (int i = 0; i < 8; ++i) printf("%u %u\n", h[i] >> 16, h[16 * 256 + i] >> 16);
    uint64_t s = 0, s2 = 0;
    for (int i = 0; i < 16 * 256; ++i) { s += h[i] >> 16; s2 += h[16 * 256 + i] >> 16; }
    printf("sum %llu %llu\n", (unsigned long long)s, (unsigned long long)s2);
    return 0;
}
