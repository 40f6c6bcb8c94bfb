#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
// Does a same-address LDS atomic with return hand out its old values in ascending LANE order?
__global__ void k(int pattern, int *out_bad, int *dump)
{
    __shared__ int slot[64];
    const int lane = threadIdx.x & 63;
    for (int rep = 0; rep < 64; ++rep) {
        slot[lane] = 0;
        __builtin_amdgcn_wave_barrier();
        int key;
        switch (pattern) {
        case 0: key = 0; break;                       // all lanes one address
        case 1: key = lane & 3; break;                // 4 addresses interleaved
        case 2: key = lane >> 4; break;               // 4 addresses, blocks of 16
        case 3: key = (lane * 7 + rep) % 5; break;    // irregular
        case 4: key = (lane ^ rep) & 7; break;
        default: key = ((lane * 2654435761u) >> 27) % (1 + rep % 9); break;
        }
        const bool act = pattern < 6 ? true : ((lane * 13 + rep) % 3 != 0);    // pattern 6: some lanes inactive
        int old = -1;
        if (act) old = atomicAdd(&slot[key], 1);
        __builtin_amdgcn_wave_barrier();
        // expected: number of ACTIVE lanes below me with the same key
        int expect = 0;
        for (int l = 0; l < 64; ++l) {
            const int kl = __shfl(key, l, 64);
            const int al = __shfl((int)act, l, 64);
            if (l < lane && al && kl == key) ++expect;
        }
        if (act && old != expect) atomicAdd(out_bad, 1);
        if (rep == 3 && pattern == 5) dump[lane] = old * 100 + key;
    }
}
int main()
{
    int *d_bad, *d_dump;
    hipMalloc(&d_bad, 4); hipMalloc(&d_dump, 256);
    for (int p = 0; p < 7; ++p) {
        hipMemset(d_bad, 0, 4);
        k<<<64, 64>>>(p, d_bad, d_dump);
        int bad = -1;
        hipMemcpy(&bad, d_bad, 4, hipMemcpyDeviceToHost);
        printf("pattern %d: %d lanes with old != rank-in-lane-order (of %d)\n", p, bad, 64 * 64 * 64);
    }
    return 0;
}
