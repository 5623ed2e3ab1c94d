// Compile-time generated small DFT codelets for the timeslot axis (M = 2..31 ...) and helpers.
//
// Everything is expressed with static (template) loops so that every array index and every twiddle
// factor is a constant expression: the data stays in VGPRs and the twiddles become literals.
// Twiddles come from a constexpr sine/cosine evaluated in double with exact octant reduction on the
// rational angle e/n, so W^0, W^(n/4), W^(n/2) ... are exactly 1, -j, -1 ...
#pragma once
#ifndef __HIPCC_RTC__
#include <hip/hip_runtime.h>
#include <stdint.h>
#else
typedef signed long long int64_t;
#endif
#include <type_traits>

namespace gfdm {
namespace dft {

typedef float2 cf;

template <int B, int E, class F>
__device__ __forceinline__ void static_for(F&& f)
{
    if constexpr (B < E) {
        f(std::integral_constant<int, B>{});
        static_for<B + 1, E>(f);
    }
}

// ---------------------------------------------------------------- constexpr trigonometry
constexpr double kPi = 3.141592653589793238462643383279502884;

constexpr double sin_small(double x)      // |x| <= pi/4
{
    double x2 = x * x, term = x, sum = x;
    for (int i = 1; i < 14; ++i) { term *= -x2 / (double)((2 * i) * (2 * i + 1)); sum += term; }
    return sum;
}
constexpr double cos_small(double x)
{
    double x2 = x * x, term = 1.0, sum = 1.0;
    for (int i = 1; i < 14; ++i) { term *= -x2 / (double)((2 * i - 1) * (2 * i)); sum += term; }
    return sum;
}
// cos / sin of 2*pi*e/n
constexpr double cos2pi(long e, long n)
{
    const long full = 8 * n, half = 4 * n, quarter = 2 * n, eighth = n;
    long num = (8 * (e % n)) % full;
    if (num < 0) num += full;
    double sign = 1.0;
    if (num > half) num = full - num;
    if (num > quarter) { num = half - num; sign = -1.0; }
    const double r = (num <= eighth) ? cos_small(2.0 * kPi * (double)num / (double)full)
                                     : sin_small(2.0 * kPi * (double)(quarter - num) / (double)full);
    return sign * r;
}
constexpr double sin2pi(long e, long n)
{
    const long full = 8 * n, half = 4 * n, quarter = 2 * n, eighth = n;
    long num = (8 * (e % n)) % full;
    if (num < 0) num += full;
    double sign = 1.0;
    if (num > half) { num = full - num; sign = -1.0; }
    if (num > quarter) num = half - num;
    const double r = (num <= eighth) ? sin_small(2.0 * kPi * (double)num / (double)full)
                                     : cos_small(2.0 * kPi * (double)(quarter - num) / (double)full);
    return sign * r;
}

// ---------------------------------------------------------------- complex helpers
__device__ __forceinline__ cf mk(float a, float b) { return make_float2(a, b); }
__device__ __forceinline__ cf operator+(cf a, cf b) { return mk(a.x + b.x, a.y + b.y); }
__device__ __forceinline__ cf operator-(cf a, cf b) { return mk(a.x - b.x, a.y - b.y); }
__device__ __forceinline__ cf cmul(cf a, cf b) { return mk(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x); }
__device__ __forceinline__ cf cmulc(cf a, cf b) { return mk(a.x * b.x + a.y * b.y, a.y * b.x - a.x * b.y); }   // a * conj(b)
__device__ __forceinline__ cf cfma(cf a, cf b, cf c) { return mk(fmaf(a.x, b.x, fmaf(-a.y, b.y, c.x)), fmaf(a.x, b.y, fmaf(a.y, b.x, c.y))); }
__device__ __forceinline__ cf scale(cf a, float s) { return mk(a.x * s, a.y * s); }
// Streaming accesses to the block data: every sample is read once and every result written once, so these loads and stores carry
// the non-temporal hint (`nt`).  Measured on MI355X (K=64 M=9, rocprofv3 kernel time): with plain stores the written lines sit
// dirty in L2 and drain at the END of the kernel -- a 4096-block launch (18.9 MB written) took 12.2 us, with nt stores 10.9 us
// (modulate 12.2 -> 10.9, ZF + 2 IC 17.0 -> 15.2); nt loads add 2-4 % at 65 536 blocks.  Explicit scope bits (sc0 / sc1 through
// a raw buffer store) measured no better than nt alone.
typedef float v2f_io __attribute__((ext_vector_type(2)));
__device__ __forceinline__ cf ld_stream(const cf* p)
{
    const v2f_io v = __builtin_nontemporal_load(reinterpret_cast<const v2f_io*>(p));
    return mk(v.x, v.y);
}
// base: buffer pointer, idx: element index of this lane
__device__ __forceinline__ void st_stream(cf* base, int64_t idx, cf v)
{
    __builtin_nontemporal_store(v2f_io{ v.x, v.y }, reinterpret_cast<v2f_io*>(base + idx));
}

template <bool INV> __device__ __forceinline__ cf mul_mj(cf a) { return INV ? mk(-a.y, a.x) : mk(a.y, -a.x); }     // * (-j) forward, * (+j) inverse
template <bool INV> __device__ __forceinline__ cf cmul_dir(cf a, cf w) { return INV ? cmulc(a, w) : cmul(a, w); }   // w is a FORWARD root

// a * exp(-+ 2 pi j E / N) with the factor folded at compile time
template <int E, int N, bool INV>
__device__ __forceinline__ cf mul_root(cf a)
{
    constexpr int e = ((E % N) + N) % N;
    if constexpr (e == 0) {
        return a;
    } else if constexpr (2 * e == N) {
        return mk(-a.x, -a.y);
    } else if constexpr (4 * e == N) {
        return mul_mj<INV>(a);
    } else if constexpr (4 * e == 3 * N) {
        return mul_mj<!INV>(a);
    } else {
        constexpr float c = (float)cos2pi(e, N);
        constexpr float s = (float)(INV ? sin2pi(e, N) : -sin2pi(e, N));
        return mk(fmaf(a.x, c, -a.y * s), fmaf(a.x, s, a.y * c));
    }
}

constexpr int smallest_factor(int n)
{
    if (n % 4 == 0) return 4;
    for (int f = 2; f * f <= n; ++f)
        if (n % f == 0) return f;
    return n;
}

// ---------------------------------------------------------------- codelets
template <int N, bool INV> struct Dft;

template <bool INV> struct Dft<1, INV> { static __device__ __forceinline__ void run(cf (&)[1]) {} };

template <bool INV> struct Dft<2, INV> {
    static __device__ __forceinline__ void run(cf (&x)[2]) { const cf a = x[0], b = x[1]; x[0] = a + b; x[1] = a - b; }
};

template <bool INV> struct Dft<4, INV> {
    static __device__ __forceinline__ void run(cf (&x)[4])
    {
        const cf apc = x[0] + x[2], amc = x[0] - x[2], bpd = x[1] + x[3], bmd = mul_mj<INV>(x[1] - x[3]);
        x[0] = apc + bpd; x[1] = amc + bmd; x[2] = apc - bpd; x[3] = amc - bmd;
    }
};

// odd prime P: symmetric direct form, (P-1)^2/2 real multiply-adds per component pair
template <int P, bool INV> struct PrimeDft {
    static constexpr int H = (P - 1) / 2;
    static __device__ __forceinline__ void run(cf (&x)[P])
    {
        cf a[H], b[H];
        static_for<0, H>([&](auto i) { constexpr int n = decltype(i)::value + 1; a[n - 1] = x[n] + x[P - n]; b[n - 1] = x[n] - x[P - n]; });
        const cf x0 = x[0];
        cf sum = x0;
        static_for<0, H>([&](auto i) { sum = sum + a[decltype(i)::value]; });
        x[0] = sum;
        if constexpr (P >= 11) {
            // long accumulations: both components of a complex term share the real factor, i.e. one packed v_pk_fma_f32 per
            // term (the build disables automatic SLP packing, -fno-slp-vectorize; here the pairing is free of shuffles)
            typedef float v2f __attribute__((ext_vector_type(2)));
            v2f av[H], bv[H];
            static_for<0, H>([&](auto i) { constexpr int n = decltype(i)::value; av[n] = v2f{ a[n].x, a[n].y }; bv[n] = v2f{ b[n].x, b[n].y }; });
            static_for<0, H>([&](auto ki) {
                constexpr int k = decltype(ki)::value + 1;
                v2f re = v2f{ x0.x, x0.y }, im = v2f{ 0.f, 0.f };
                static_for<0, H>([&](auto ni) {
                    constexpr int n = decltype(ni)::value + 1;
                    constexpr float c = (float)cos2pi((long)n * k, P);
                    constexpr float s = (float)sin2pi((long)n * k, P);
                    re = __builtin_elementwise_fma(av[n - 1], v2f{ c, c }, re);
                    im = __builtin_elementwise_fma(bv[n - 1], v2f{ s, s }, im);
                });
                const cf lo = mk(re.x + im.y, re.y - im.x), hi = mk(re.x - im.y, re.y + im.x);
                x[k] = INV ? hi : lo;
                x[P - k] = INV ? lo : hi;
            });
            return;
        }
        static_for<0, H>([&](auto ki) {
            constexpr int k = decltype(ki)::value + 1;
            cf re = x0, im = mk(0.f, 0.f);
            static_for<0, H>([&](auto ni) {
                constexpr int n = decltype(ni)::value + 1;
                constexpr float c = (float)cos2pi((long)n * k, P);
                constexpr float s = (float)sin2pi((long)n * k, P);
                re = mk(fmaf(a[n - 1].x, c, re.x), fmaf(a[n - 1].y, c, re.y));
                im = mk(fmaf(b[n - 1].x, s, im.x), fmaf(b[n - 1].y, s, im.y));
            });
            // forward: y[k] = re - j im, y[P-k] = re + j im ; inverse swaps them
            const cf lo = mk(re.x + im.y, re.y - im.x), hi = mk(re.x - im.y, re.y + im.x);
            x[k] = INV ? hi : lo;
            x[P - k] = INV ? lo : hi;
        });
    }
};

constexpr int gcd_of(int a, int b) { return b == 0 ? a : gcd_of(b, a % b); }
constexpr int inverse_mod(int a, int m)                                       // a^-1 mod m (a, m coprime)
{
    for (int i = 1; i < m; ++i)
        if ((a * i) % m == 1) return i;
    return 1;
}

// general N: prime -> direct; N = R S with coprime factors -> prime-factor algorithm (Good-Thomas: the index maps
// n = (S n1 + R n2) mod N, k = (S S' k1 + R R' k2) mod N with S S' = 1 mod R, R R' = 1 mod S turn the N-point transform into an R x S
// two-dimensional one WITHOUT twiddle factors -- 15 = 3 x 5 saves eight complex multiplies, and the maps are compile-time
// register renaming); otherwise Cooley-Tukey  n = s + S r,  k = k1 + R k2
template <int N, bool INV> struct Dft {
    static constexpr int R = smallest_factor(N);
    static constexpr int S = N / R;
    static __device__ __forceinline__ void run(cf (&x)[N])
    {
        if constexpr (R == N) {
            PrimeDft<N, INV>::run(x);
        } else if constexpr (gcd_of(R, S) == 1) {
            constexpr int Si = inverse_mod(S % R, R), Ri = inverse_mod(R % S, S);
            cf y[N];                                   // y[n2 * R + k1]
            static_for<0, S>([&](auto si) {
                constexpr int n2 = decltype(si)::value;
                cf t[R];
                static_for<0, R>([&](auto ri) { constexpr int n1 = decltype(ri)::value; t[n1] = x[(S * n1 + R * n2) % N]; });
                Dft<R, INV>::run(t);
                static_for<0, R>([&](auto ki) { constexpr int k1 = decltype(ki)::value; y[n2 * R + k1] = t[k1]; });
            });
            static_for<0, R>([&](auto ki) {
                constexpr int k1 = decltype(ki)::value;
                cf t[S];
                static_for<0, S>([&](auto si) { constexpr int n2 = decltype(si)::value; t[n2] = y[n2 * R + k1]; });
                Dft<S, INV>::run(t);
                static_for<0, S>([&](auto k2i) { constexpr int k2 = decltype(k2i)::value; x[(S * Si * k1 + R * Ri * k2) % N] = t[k2]; });
            });
        } else {
            cf y[N];                                   // y[s*R + k1]
            static_for<0, S>([&](auto si) {
                constexpr int s = decltype(si)::value;
                cf t[R];
                static_for<0, R>([&](auto ri) { constexpr int r = decltype(ri)::value; t[r] = x[s + S * r]; });
                Dft<R, INV>::run(t);
                static_for<0, R>([&](auto ki) { constexpr int k1 = decltype(ki)::value; y[s * R + k1] = mul_root<s * k1, N, INV>(t[k1]); });
            });
            static_for<0, R>([&](auto ki) {
                constexpr int k1 = decltype(ki)::value;
                cf t[S];
                static_for<0, S>([&](auto si) { constexpr int s = decltype(si)::value; t[s] = y[s * R + k1]; });
                Dft<S, INV>::run(t);
                static_for<0, S>([&](auto k2i) { constexpr int k2 = decltype(k2i)::value; x[k1 + R * k2] = t[k2]; });
            });
        }
    }
};

template <int N, bool INV>
__device__ __forceinline__ void dft_inplace(cf (&x)[N]) { Dft<N, INV>::run(x); }

}  // namespace dft
}  // namespace gfdm
