#include "gfdm_rowlane_impl.h"
#include <cstdio>
using namespace gfdm;
__global__ void k(float* out)
{
    const int lane = threadIdx.x;
    float a0 = lane * 4 + 0, a1 = lane * 4 + 1, a2 = lane * 4 + 2, a3 = lane * 4 + 3;
    lane_row_transpose4(a0, a1, a2, a3);
    out[lane * 4 + 0] = a0; out[lane * 4 + 1] = a1; out[lane * 4 + 2] = a2; out[lane * 4 + 3] = a3;
}
int main()
{
    float* d; (void)hipMalloc(&d, 256 * 4);
    k<<<1, 64>>>(d);
    float h[256]; (void)hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int lane = 0; lane < 64; ++lane)
        for (int i = 0; i < 4; ++i) {
            const int rr = lane >> 4, tq = lane & 15;
            const float expect = (16 * i + tq) * 4 + rr;       // register rr of lane (i, tq)
            if (h[lane * 4 + i] != expect) { if (bad < 8) printf("lane %d reg %d: got %g expect %g\n", lane, i, h[lane * 4 + i], expect); ++bad; }
        }
    printf("bad = %d\n", bad);
    return 0;
}
