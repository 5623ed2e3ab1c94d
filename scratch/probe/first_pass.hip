// Probe: wave_fft_first_pass<9> against a host reference of pass 0.   hipcc --offload-arch=gfx950 -I../../gr-gfdm_amd/csrc first_pass.hip
#include "gfdm_rowlane_impl.h"
#include <cstdio>
#include <complex>
#include <vector>
using namespace gfdm;
__global__ void k(cf* out, const cf* in, const cf* wK)
{
    __shared__ cf tile[64 * 9];
    const int q = threadIdx.x;
    cf row[9];
    for (int m = 0; m < 9; ++m) row[m] = in[q * 9 + m];
    FftTwiddles<64> twd;
    load_fft_twiddles<64>(twd, q, wK);
    wave_fft_first_pass<9, false>(tile, q, twd, row);
    for (int m = 0; m < 9; ++m) out[q * 9 + m] = tile[FftLayout<64>::slot(q) * 9 + m];     // row q in natural order
}
int main()
{
    const int K = 64, M = 9;
    std::vector<std::complex<float>> x(K * M), w(K), got(K * M);
    for (int i = 0; i < K * M; ++i) x[i] = { (float)((i * 37) % 101) / 50.f - 1.f, (float)((i * 53) % 89) / 40.f - 1.f };
    for (int i = 0; i < K; ++i) w[i] = std::polar(1.0f, (float)(-2.0 * M_PI * i / K));
    cf *din, *dout, *dw;
    (void)hipMalloc(&din, K * M * 8); (void)hipMalloc(&dout, K * M * 8); (void)hipMalloc(&dw, K * 8);
    (void)hipMemcpy(din, x.data(), K * M * 8, hipMemcpyHostToDevice); (void)hipMemcpy(dw, w.data(), K * 8, hipMemcpyHostToDevice);
    k<<<1, 64>>>(dout, din, dw);
    (void)hipMemcpy(got.data(), dout, K * M * 8, hipMemcpyDeviceToHost);
    double worst = 0; int wr = -1, wc = -1;
    for (int tq = 0; tq < 16; ++tq)
        for (int m = 0; m < M; ++m)
            for (int u = 0; u < 4; ++u) {
                std::complex<double> y = 0;
                for (int r = 0; r < 4; ++r) y += std::complex<double>(x[(tq + 16 * r) * M + m]) * std::polar(1.0, -2.0 * M_PI * r * u / 4);
                y *= std::polar(1.0, -2.0 * M_PI * tq * u / K);
                double e = std::abs(y - std::complex<double>(got[(4 * tq + u) * M + m]));
                if (e > worst) { worst = e; wr = 4 * tq + u; wc = m; }
            }
    printf("worst error %.3g at row %d col %d\n", worst, wr, wc);
    return 0;
}
