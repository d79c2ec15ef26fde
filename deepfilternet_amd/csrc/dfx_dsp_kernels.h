// HIP kernels for the libDF half of enhance(): batched STFT analysis (+ERB dB features), exponential-norm scans,
// ISTFT + overlap-add, ERB filterbank ops and the fused deep-filter / ERB-gain application.
//
// All of these are HBM-bound: the design goal is coalesced 4..16-byte accesses, frames staged through
// LDS, one wave (64 lanes) per frame FFT, and every byte touched once.  No MFMA here on purpose.
//
// Reference semantics (each kernel cites its lines): libDF/src/lib.rs, libDF/src/transforms.rs, pyDF/src/lib.rs,
// DeepFilterNet/df/multiframe.py, DeepFilterNet/df/modules.py, DeepFilterNet/df/deepfilternet3.py.
#pragma once

#include "dfx_common.h"

#define DFX_DSP_TEAM 64    // lanes per frame (one wave)
#define DFX_DSP_TEAMS 8    // frames in flight per workgroup
#define DFX_DSP_THREADS (DFX_DSP_TEAM * DFX_DSP_TEAMS)

static __device__ __forceinline__ float2 dfx_cmul(float2 a, float2 b) {
    return make_float2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x);
}
static __device__ __forceinline__ float2 dfx_cadd(float2 a, float2 b) { return make_float2(a.x + b.x, a.y + b.y); }
static __device__ __forceinline__ float2 dfx_csub(float2 a, float2 b) { return make_float2(a.x - b.x, a.y - b.y); }
// multiply by -i (forward, SG=-1) or +i (inverse, SG=+1)
template <int SG>
static __device__ __forceinline__ float2 dfx_mul_sgi(float2 a) {
    return SG < 0 ? make_float2(a.y, -a.x) : make_float2(-a.y, a.x);
}

// One Stockham autosort pass of radix R over a length-M complex sequence held in LDS.
//   y[q + s*(R*p + j)] = w^(j*p) * DFT_R(x[q + s*(p + m*j)])_j ,  p < m = ncur/R, q < s,  w = exp(SG*2*pi*i/ncur)
// tw is the length-N table exp(-2*pi*i*k/N), N = 2*M (so w^(j*p) = tw[j*p*(N/ncur)], conjugated for the inverse).
template <int R, int SG>
static __device__ __forceinline__ void dfx_fft_pass(const float2 *x, float2 *y, const float2 *tw, int M, int N, int ncur,
                                                    int s, int lane) {
    const int m = ncur / R;
    const int nbf = M / R;  // butterflies in this pass
    const int tws = N / ncur;
    for (int b = lane; b < nbf; b += DFX_DSP_TEAM) {
        const int p = b / s, q = b - p * s;
        float2 a[R];
#pragma unroll
        for (int j = 0; j < R; ++j) a[j] = x[q + s * (p + m * j)];
        float2 o[R];
        if constexpr (R == 2) {
            o[0] = dfx_cadd(a[0], a[1]);
            o[1] = dfx_csub(a[0], a[1]);
        } else if constexpr (R == 3) {
            const float2 sm = dfx_cadd(a[1], a[2]), d = dfx_csub(a[1], a[2]);
            const float2 mm = make_float2(a[0].x - 0.5f * sm.x, a[0].y - 0.5f * sm.y);
            float2 jd = dfx_mul_sgi<SG>(d);
            jd.x *= 0.86602540378443864676f;
            jd.y *= 0.86602540378443864676f;
            o[0] = dfx_cadd(a[0], sm);
            o[1] = dfx_cadd(mm, jd);
            o[2] = dfx_csub(mm, jd);
        } else if constexpr (R == 4) {
            const float2 t0 = dfx_cadd(a[0], a[2]), t1 = dfx_csub(a[0], a[2]);
            const float2 t2 = dfx_cadd(a[1], a[3]), t3 = dfx_mul_sgi<SG>(dfx_csub(a[1], a[3]));
            o[0] = dfx_cadd(t0, t2);
            o[1] = dfx_cadd(t1, t3);
            o[2] = dfx_csub(t0, t2);
            o[3] = dfx_csub(t1, t3);
        } else {  // R == 5
            const float c1 = 0.30901699437494742410f, s1 = 0.95105651629515357212f;
            const float c2 = -0.80901699437494742410f, s2 = 0.58778525229247312917f;
            const float2 s14 = dfx_cadd(a[1], a[4]), d14 = dfx_csub(a[1], a[4]);
            const float2 s23 = dfx_cadd(a[2], a[3]), d23 = dfx_csub(a[2], a[3]);
            const float2 m1 = make_float2(a[0].x + c1 * s14.x + c2 * s23.x, a[0].y + c1 * s14.y + c2 * s23.y);
            const float2 m2 = make_float2(a[0].x + c2 * s14.x + c1 * s23.x, a[0].y + c2 * s14.y + c1 * s23.y);
            const float2 n1 = dfx_mul_sgi<SG>(make_float2(s1 * d14.x + s2 * d23.x, s1 * d14.y + s2 * d23.y));
            const float2 n2 = dfx_mul_sgi<SG>(make_float2(s2 * d14.x - s1 * d23.x, s2 * d14.y - s1 * d23.y));
            o[0] = make_float2(a[0].x + s14.x + s23.x, a[0].y + s14.y + s23.y);
            o[1] = dfx_cadd(m1, n1);
            o[4] = dfx_csub(m1, n1);
            o[2] = dfx_cadd(m2, n2);
            o[3] = dfx_csub(m2, n2);
        }
        y[q + s * (R * p)] = o[0];
#pragma unroll
        for (int j = 1; j < R; ++j) {
            float2 w = tw[j * p * tws];
            if (SG > 0) w.y = -w.y;
            y[q + s * (R * p + j)] = dfx_cmul(o[j], w);
        }
    }
}

// The same pass with every size a compile-time constant: the butterfly index split b -> (p, q), the strides and the twiddle step
// become shifts / multiplies by constants.  The generic pass spends most of its instructions on that index arithmetic (the
// kernels were VALU-issue bound at ~2500 instructions per frame); the arithmetic on the data is unchanged, so the bits are too.
template <int R, int SG, int M, int NCUR, int S>
static __device__ __forceinline__ void dfx_fft_pass_c(const float2 *x, float2 *y, const float2 *tw, int lane) {
    constexpr int m = NCUR / R, nbf = M / R, tws = (2 * M) / NCUR;
    for (int b = lane; b < nbf; b += DFX_DSP_TEAM) {
        const int p = b / S, q = b - p * S;
        float2 a[R];
#pragma unroll
        for (int j = 0; j < R; ++j) a[j] = x[q + S * (p + m * j)];
        float2 o[R];
        if constexpr (R == 2) {
            o[0] = dfx_cadd(a[0], a[1]);
            o[1] = dfx_csub(a[0], a[1]);
        } else if constexpr (R == 3) {
            const float2 sm = dfx_cadd(a[1], a[2]), d = dfx_csub(a[1], a[2]);
            const float2 mm = make_float2(a[0].x - 0.5f * sm.x, a[0].y - 0.5f * sm.y);
            float2 jd = dfx_mul_sgi<SG>(d);
            jd.x *= 0.86602540378443864676f;
            jd.y *= 0.86602540378443864676f;
            o[0] = dfx_cadd(a[0], sm);
            o[1] = dfx_cadd(mm, jd);
            o[2] = dfx_csub(mm, jd);
        } else if constexpr (R == 4) {
            const float2 t0 = dfx_cadd(a[0], a[2]), t1 = dfx_csub(a[0], a[2]);
            const float2 t2 = dfx_cadd(a[1], a[3]), t3 = dfx_mul_sgi<SG>(dfx_csub(a[1], a[3]));
            o[0] = dfx_cadd(t0, t2);
            o[1] = dfx_cadd(t1, t3);
            o[2] = dfx_csub(t0, t2);
            o[3] = dfx_csub(t1, t3);
        } else {  // R == 5
            const float c1 = 0.30901699437494742410f, s1 = 0.95105651629515357212f;
            const float c2 = -0.80901699437494742410f, s2 = 0.58778525229247312917f;
            const float2 s14 = dfx_cadd(a[1], a[4]), d14 = dfx_csub(a[1], a[4]);
            const float2 s23 = dfx_cadd(a[2], a[3]), d23 = dfx_csub(a[2], a[3]);
            const float2 m1 = make_float2(a[0].x + c1 * s14.x + c2 * s23.x, a[0].y + c1 * s14.y + c2 * s23.y);
            const float2 m2 = make_float2(a[0].x + c2 * s14.x + c1 * s23.x, a[0].y + c2 * s14.y + c1 * s23.y);
            const float2 n1 = dfx_mul_sgi<SG>(make_float2(s1 * d14.x + s2 * d23.x, s1 * d14.y + s2 * d23.y));
            const float2 n2 = dfx_mul_sgi<SG>(make_float2(s2 * d14.x - s1 * d23.x, s2 * d14.y - s1 * d23.y));
            o[0] = make_float2(a[0].x + s14.x + s23.x, a[0].y + s14.y + s23.y);
            o[1] = dfx_cadd(m1, n1);
            o[4] = dfx_csub(m1, n1);
            o[2] = dfx_cadd(m2, n2);
            o[3] = dfx_csub(m2, n2);
        }
        y[q + S * (R * p)] = o[0];
#pragma unroll
        for (int j = 1; j < R; ++j) {
            float2 w = tw[j * p * tws];
            if (SG > 0) w.y = -w.y;
            y[q + S * (R * p + j)] = dfx_cmul(o[j], w);
        }
    }
}

// multiply by exp(SG * 2*pi*i * K / 8), K = 1 or 3 (the two 8th roots that are not axis rotations)
template <int SG, int K>
static __device__ __forceinline__ float2 dfx_mul_w8(float2 a) {
    const float h = 0.70710678118654752440f;
    if constexpr (K == 1) return SG < 0 ? make_float2((a.x + a.y) * h, (a.y - a.x) * h) : make_float2((a.x - a.y) * h, (a.y + a.x) * h);
    else return SG < 0 ? make_float2((a.y - a.x) * h, (-a.x - a.y) * h) : make_float2((-a.x - a.y) * h, (a.x - a.y) * h);
}
// 4- and 5-point transforms on values (the butterflies of the passes below)
template <int SG>
static __device__ __forceinline__ void dfx_dft4(float2 x0, float2 x1, float2 x2, float2 x3, float2 &y0, float2 &y1, float2 &y2, float2 &y3) {
    const float2 t0 = dfx_cadd(x0, x2), t1 = dfx_csub(x0, x2);
    const float2 t2 = dfx_cadd(x1, x3), t3 = dfx_mul_sgi<SG>(dfx_csub(x1, x3));
    y0 = dfx_cadd(t0, t2);
    y1 = dfx_cadd(t1, t3);
    y2 = dfx_csub(t0, t2);
    y3 = dfx_csub(t1, t3);
}
template <int SG>
static __device__ __forceinline__ void dfx_dft5(float2 x0, float2 x1, float2 x2, float2 x3, float2 x4, float2 &y0, float2 &y1, float2 &y2,
                                                float2 &y3, float2 &y4) {
    const float c1 = 0.30901699437494742410f, s1 = 0.95105651629515357212f;
    const float c2 = -0.80901699437494742410f, s2 = 0.58778525229247312917f;
    const float2 s14 = dfx_cadd(x1, x4), d14 = dfx_csub(x1, x4);
    const float2 s23 = dfx_cadd(x2, x3), d23 = dfx_csub(x2, x3);
    const float2 m1 = make_float2(x0.x + c1 * s14.x + c2 * s23.x, x0.y + c1 * s14.y + c2 * s23.y);
    const float2 m2 = make_float2(x0.x + c2 * s14.x + c1 * s23.x, x0.y + c2 * s14.y + c1 * s23.y);
    const float2 n1 = dfx_mul_sgi<SG>(make_float2(s1 * d14.x + s2 * d23.x, s1 * d14.y + s2 * d23.y));
    const float2 n2 = dfx_mul_sgi<SG>(make_float2(s2 * d14.x - s1 * d23.x, s2 * d14.y - s1 * d23.y));
    y0 = make_float2(x0.x + s14.x + s23.x, x0.y + s14.y + s23.y);
    y1 = dfx_cadd(m1, n1);
    y4 = dfx_csub(m1, n1);
    y2 = dfx_cadd(m2, n2);
    y3 = dfx_csub(m2, n2);
}

// 8-, 6- and 10-point transforms on values: the butterflies of the 8 x 6 x 10 plan (dfx_fft_pass_ip, dfx_fft480_t)
template <int SG>
static __device__ __forceinline__ void dfx_bfly8(const float2 *a, float2 *o) {
    // even / odd inputs through two 4-point transforms, then o[k] = E[k] + W8^k O[k], o[k + 4] = E[k] - W8^k O[k]
    float2 e0, e1, e2, e3, q0, q1, q2, q3;
    dfx_dft4<SG>(a[0], a[2], a[4], a[6], e0, e1, e2, e3);
    dfx_dft4<SG>(a[1], a[3], a[5], a[7], q0, q1, q2, q3);
    q1 = dfx_mul_w8<SG, 1>(q1);
    q2 = dfx_mul_sgi<SG>(q2);
    q3 = dfx_mul_w8<SG, 3>(q3);
    o[0] = dfx_cadd(e0, q0), o[4] = dfx_csub(e0, q0);
    o[1] = dfx_cadd(e1, q1), o[5] = dfx_csub(e1, q1);
    o[2] = dfx_cadd(e2, q2), o[6] = dfx_csub(e2, q2);
    o[3] = dfx_cadd(e3, q3), o[7] = dfx_csub(e3, q3);
}
template <int SG>
static __device__ __forceinline__ void dfx_bfly10(const float2 *a, float2 *o) {
    // 10 = 2 x 5 with coprime factors: input n = 5 n1 + 2 n2, output k = 5 k1 + 6 k2 (mod 10), no twiddles in between
    dfx_dft5<SG>(dfx_cadd(a[0], a[5]), dfx_cadd(a[2], a[7]), dfx_cadd(a[4], a[9]), dfx_cadd(a[6], a[1]), dfx_cadd(a[8], a[3]), o[0], o[6], o[2], o[8], o[4]);
    dfx_dft5<SG>(dfx_csub(a[0], a[5]), dfx_csub(a[2], a[7]), dfx_csub(a[4], a[9]), dfx_csub(a[6], a[1]), dfx_csub(a[8], a[3]), o[5], o[1], o[7], o[3], o[9]);
}
template <int SG>
static __device__ __forceinline__ void dfx_bfly6(const float2 *a, float2 *o) {
    // 6 = 2 x 3 with coprime factors: input n = 3 n1 + 2 n2, output k = 3 k1 + 4 k2 (mod 6) need no twiddles between the
    // three 2-point and the two 3-point transforms
    const float2 s0 = dfx_cadd(a[0], a[3]), t0 = dfx_csub(a[0], a[3]);
    const float2 s1 = dfx_cadd(a[2], a[5]), t1 = dfx_csub(a[2], a[5]);
    const float2 s2 = dfx_cadd(a[4], a[1]), t2 = dfx_csub(a[4], a[1]);
    auto dft3 = [](float2 x0, float2 x1, float2 x2, float2 &y0, float2 &y1, float2 &y2) {
        const float2 sm = dfx_cadd(x1, x2), d = dfx_csub(x1, x2);
        const float2 mm = make_float2(x0.x - 0.5f * sm.x, x0.y - 0.5f * sm.y);
        float2 jd = dfx_mul_sgi<SG>(d);
        jd.x *= 0.86602540378443864676f;
        jd.y *= 0.86602540378443864676f;
        y0 = dfx_cadd(x0, sm);
        y1 = dfx_cadd(mm, jd);
        y2 = dfx_csub(mm, jd);
    };
    dft3(s0, s1, s2, o[0], o[4], o[2]);   // k1 = 0: k = 4 k2 mod 6
    dft3(t0, t1, t2, o[3], o[1], o[5]);   // k1 = 1: k = 3 + 4 k2 mod 6
}

// The same pass IN PLACE: a lane first reads the inputs of all its butterflies, then writes their outputs.  One team is one wave, whose
// lanes run in lockstep and whose LDS accesses complete in program order, so every read of the pass precedes every write (the wave-level
// sync between the two halves costs nothing on the GPU and is what the CPU interpreter needs).  Same butterflies, same twiddles, same
// order of operations as dfx_fft_pass_c — same bits — with half the LDS per frame: 3 workgroups per CU instead of 2 (the STFT kernels
// wait on LDS / memory latencies for more than half of their wave-cycles; profiles/r02_pmc_stft_kernels.txt).
// PIN / POUT (round 5): the pass reads / writes the PADDED layout, element i at slot i + (i >> 3) (one float2 of padding per eight).  The autosort
// passes write with a stride of S * R... elements: 8 float2 = 64 bytes in the first pass of the 8 x 6 x 10 plan — every lane of a ds_write_b64 in one
// of two banks, a 32-way conflict — and 48 float2 in the second (8-way).  The counters of the round-4 kernels (profiles/
// r05_pmc_side_kernels_start_of_round.txt) show the STFT kernels' LDS pipe 82 % busy with half of its active cycles bank conflicts.  Since a pass
// reads everything before it writes anything, the layout may change inside a pass: the first pass reads the natural layout and writes the padded
// one, the last reads padded and writes natural (unit stride either way) — nothing outside the transform sees the padding; the frame buffer is
// DFX_FFT480_BUF slots long.  Every slot index stays `lane-dependent base + compile-time offset` (the j-dependent parts are multiples of 8 or are
// folded into one base per residue), so the passes cost no extra address arithmetic per access.
#define DFX_FFT480_BUF 544   /* float2 slots per in-place frame buffer: 480 + 59 of padding, rounded up to a multiple of 4 (16-byte carve) */
template <int C>
static __device__ __forceinline__ int dfx_pad_idx(int i0, int j) {   // slot of element i0 + C * j (C * j known after unrolling)
    const int cj = C * j, rem = cj & 7;
    return i0 + ((i0 + rem) >> 3) + cj + ((cj - rem) >> 3);
}
template <int R, int SG, int M, int NCUR, int S, bool PIN = false, bool POUT = false>
static __device__ __forceinline__ void dfx_fft_pass_ip(float2 *x, const float2 *tw, int lane, bool active) {
    constexpr int m = NCUR / R, nbf = M / R, tws = (2 * M) / NCUR, NR = (nbf + DFX_DSP_TEAM - 1) / DFX_DSP_TEAM;
    float2 a[NR][R];
    // the twiddles are read here too, with the data: the table lives in the same LDS allocation as the frames, so a read of it placed
    // between the stores of the second half cannot be moved across them by the compiler and every one of them became a separate LDS
    // round trip (load, wait, multiply, store: 6-8 dependent latencies per pass instead of one)
    float2 w[NR][R - 1];
    if (active) {
#pragma unroll
        for (int r = 0; r < NR; ++r) {
            // (lanes beyond the last butterfly of a partial round read butterfly 0 and drop it: a predicated LDS read costs five
            // instructions — zero the destination, save / narrow / restore exec — where the plain one costs one)
            const bool full = (r + 1) * DFX_DSP_TEAM <= nbf;
            const int b0 = lane + r * DFX_DSP_TEAM, b = full || b0 < nbf ? b0 : 0;
            const int p = b / S, q = b - p * S;
#pragma unroll
            for (int j = 0; j < R; ++j) a[r][j] = PIN ? x[dfx_pad_idx<S * m>(q + S * p, j)] : x[q + S * (p + m * j)];
            if constexpr (NCUR != R) {   // (the last pass has p = 0: every twiddle is 1)
#pragma unroll
                for (int j = 1; j < R; ++j) w[r][j - 1] = tw[j * p * tws];
            }
        }
    }
    DFX_WAVE_SYNC();
    if (active) {
#pragma unroll
        for (int r = 0; r < NR; ++r) {
            const int b = lane + r * DFX_DSP_TEAM;
            if ((r + 1) * DFX_DSP_TEAM > nbf && b >= nbf) continue;
            const int p = b / S, q = b - p * S;
            float2 o[R];
            if constexpr (R == 2) {
                o[0] = dfx_cadd(a[r][0], a[r][1]);
                o[1] = dfx_csub(a[r][0], a[r][1]);
            } else if constexpr (R == 3) {
                const float2 sm = dfx_cadd(a[r][1], a[r][2]), d = dfx_csub(a[r][1], a[r][2]);
                const float2 mm = make_float2(a[r][0].x - 0.5f * sm.x, a[r][0].y - 0.5f * sm.y);
                float2 jd = dfx_mul_sgi<SG>(d);
                jd.x *= 0.86602540378443864676f;
                jd.y *= 0.86602540378443864676f;
                o[0] = dfx_cadd(a[r][0], sm);
                o[1] = dfx_cadd(mm, jd);
                o[2] = dfx_csub(mm, jd);
            } else if constexpr (R == 4) {
                const float2 t0 = dfx_cadd(a[r][0], a[r][2]), t1 = dfx_csub(a[r][0], a[r][2]);
                const float2 t2 = dfx_cadd(a[r][1], a[r][3]), t3 = dfx_mul_sgi<SG>(dfx_csub(a[r][1], a[r][3]));
                o[0] = dfx_cadd(t0, t2);
                o[1] = dfx_cadd(t1, t3);
                o[2] = dfx_csub(t0, t2);
                o[3] = dfx_csub(t1, t3);
            } else if constexpr (R == 8) {
                dfx_bfly8<SG>(a[r], o);
            } else if constexpr (R == 10) {
                dfx_bfly10<SG>(a[r], o);
            } else if constexpr (R == 6) {
                dfx_bfly6<SG>(a[r], o);
            } else {  // R == 5
                const float c1 = 0.30901699437494742410f, s1 = 0.95105651629515357212f;
                const float c2 = -0.80901699437494742410f, s2 = 0.58778525229247312917f;
                const float2 s14 = dfx_cadd(a[r][1], a[r][4]), d14 = dfx_csub(a[r][1], a[r][4]);
                const float2 s23 = dfx_cadd(a[r][2], a[r][3]), d23 = dfx_csub(a[r][2], a[r][3]);
                const float2 m1 = make_float2(a[r][0].x + c1 * s14.x + c2 * s23.x, a[r][0].y + c1 * s14.y + c2 * s23.y);
                const float2 m2 = make_float2(a[r][0].x + c2 * s14.x + c1 * s23.x, a[r][0].y + c2 * s14.y + c1 * s23.y);
                const float2 n1 = dfx_mul_sgi<SG>(make_float2(s1 * d14.x + s2 * d23.x, s1 * d14.y + s2 * d23.y));
                const float2 n2 = dfx_mul_sgi<SG>(make_float2(s2 * d14.x - s1 * d23.x, s2 * d14.y - s1 * d23.y));
                o[0] = make_float2(a[r][0].x + s14.x + s23.x, a[r][0].y + s14.y + s23.y);
                o[1] = dfx_cadd(m1, n1);
                o[4] = dfx_csub(m1, n1);
                o[2] = dfx_cadd(m2, n2);
                o[3] = dfx_csub(m2, n2);
            }
            auto slot = [&](int j) { return POUT ? dfx_pad_idx<S>(q + S * (R * p), j) : q + S * (R * p + j); };
            x[slot(0)] = o[0];
#pragma unroll
            for (int j = 1; j < R; ++j) {
                if constexpr (NCUR != R) {
                    float2 wj = w[r][j - 1];
                    if (SG > 0) wj.y = -wj.y;
                    x[slot(j)] = dfx_cmul(o[j], wj);
                } else {
                    x[slot(j)] = o[j];
                }
            }
        }
    }
    DFX_WAVE_SYNC();
}
// the 48 kHz / 20 ms configuration of every shipped model: N = 960, M = 480 = 4*4*2*3*5 (make_plan's order)
static __host__ __device__ __forceinline__ bool dfx_plan_is_480(const DfxFftPlan &pl) {
    return pl.M == 480 && pl.nstage == 5 && pl.radix[0] == 4 && pl.radix[1] == 4 && pl.radix[2] == 2 && pl.radix[3] == 3 && pl.radix[4] == 5;
}
// (the lane index is made opaque before every pass: the passes' LDS addresses depend on nothing but the lane, and a compiler that hoists
// all ~100 of them out of the kernel's frame loop — it does — pushes the kernel from 80 to 150 registers, i.e. from six waves per SIMD to three)
#ifndef DFX_FFT480_PASSES
#define DFX_FFT480_PASSES 3
#endif
template <int SG>
static __device__ __forceinline__ void dfx_fft480_ip(float2 *x, const float2 *tw, int lane, bool active) {
    int l = lane;
    DFX_OPAQUE(l);
#if DFX_FFT480_PASSES == 3     /* 480 = 8 * 6 * 10: 60 + 80 + 48 butterflies, the 6- and 10-point ones without inner twiddles (coprime factors) */
#ifndef DFX_FFT480_PAD
#define DFX_FFT480_PAD 1   /* 0: the unpadded passes (dev A/B) */
#endif
    constexpr bool PD = DFX_FFT480_PAD != 0;
    dfx_fft_pass_ip<8, SG, 480, 480, 1, false, PD>(x, tw, l, active);
    DFX_OPAQUE(l);
    dfx_fft_pass_ip<6, SG, 480, 60, 8, PD, PD>(x, tw, l, active);
    DFX_OPAQUE(l);
    dfx_fft_pass_ip<10, SG, 480, 10, 48, PD, false>(x, tw, l, active);
#else
    dfx_fft_pass_ip<4, SG, 480, 480, 1>(x, tw, l, active);
    DFX_OPAQUE(l);
    dfx_fft_pass_ip<4, SG, 480, 120, 4>(x, tw, l, active);
    DFX_OPAQUE(l);
#if DFX_FFT480_PASSES == 5     /* the plan of make_plan, pass for pass (= the two-buffer path's bits) */
    dfx_fft_pass_ip<2, SG, 480, 30, 16>(x, tw, l, active);
    DFX_OPAQUE(l);
    dfx_fft_pass_ip<3, SG, 480, 15, 32>(x, tw, l, active);
#else                          /* 480 = 4 * 4 * 6 * 5: the radix-2 and radix-3 passes as one 6-point pass */
    dfx_fft_pass_ip<6, SG, 480, 30, 16>(x, tw, l, active);
#endif
    DFX_OPAQUE(l);
    dfx_fft_pass_ip<5, SG, 480, 5, 96>(x, tw, l, active);
#endif
}

// dfx_fft480_t (round 6): the 8 x 6 x 10 plan of dfx_fft480_ip written out for the two kernels of enhance()'s batch path — same butterflies,
// same passes, same padded layout — with everything a pass needs besides its data at `lane-dependent base + constant`:
//   * the twiddles of a pass come from tables laid out FOR that pass (DfxTw480: t1[jj][lane] = the factors of outputs 2 jj + 1, 2 jj + 2 of lane's
//     butterfly in the first pass, t2[jj][p] those of the second), read with conflict-free 16-byte accesses — the generic pass computes seven and
//     ten table indices j * p * stride per frame, each a quarter-rate 32-bit multiply (the counters of round 6, profiles/r06_pmc_stft_before.txt:
//     the STFT kernels are bound by the VALU, ~890 instructions per frame of which barely half are arithmetic);
//   * element and slot indices are shifts and adds of the lane index (9 l, l + (l >> 3), q + 54 p).
// Lanes beyond a pass's last butterfly read in-range slots and drop them.  tw480_fill builds the tables from the length-960 table.
#define DFX_TW480_T1 256                      /* f32x4 entries: [4][64] */
#define DFX_TW480_T2 32                       /* f32x4 entries: [3][10] (+ 2 pad) */
#define DFX_TW480_BYTES ((DFX_TW480_T1 + DFX_TW480_T2) * 16)
static __device__ __forceinline__ void dfx_tw480_fill(f32x4 *t1, f32x4 *t2, const float2 *tw_global, int tid, int nthreads) {
    for (int i = tid; i < DFX_TW480_T1 + 30; i += nthreads) {
        int k0, k1;
        if (i < DFX_TW480_T1) {
            const int jj = i >> 6, l = i & 63;
            k0 = (2 * (2 * jj + 1) * l) % 960, k1 = jj < 3 ? (2 * (2 * jj + 2) * l) % 960 : 0;
        } else {
            const int e = i - DFX_TW480_T1, jj = e / 10, p = e - jj * 10;
            k0 = 16 * (2 * jj + 1) * p, k1 = 16 * (2 * jj + 2) * p;
        }
        const float2 a = tw_global[k0], b = tw_global[k1];
        (i < DFX_TW480_T1 ? t1[i] : t2[i - DFX_TW480_T1]) = f32x4{a.x, a.y, b.x, b.y};
    }
}
template <int SG>
static __device__ __forceinline__ float2 dfx_tw_mul(float2 o, const f32x4 w, int half) {   // o * w (forward) or o * conj(w) (inverse)
    float2 wj = half ? make_float2(w[2], w[3]) : make_float2(w[0], w[1]);
    if (SG > 0) wj.y = -wj.y;
    return dfx_cmul(o, wj);
}
template <int SG>
static __device__ __forceinline__ void dfx_fft480_t(float2 *x, const f32x4 *t1, const f32x4 *t2, int lane, bool active) {
    {   // ---- pass 1: 60 butterflies of radix 8, element l + 60 j (natural layout) -> slot 9 l + j (padded)
        int l = lane;
        DFX_OPAQUE(l);
        float2 a[8];
        f32x4 w[4];
        {   // (an idle wave reads its own buffer too: values that are defined on one path only would be zero-filled on the other)
            const float2 *rp = x + l;
#pragma unroll
            for (int j = 0; j < 8; ++j) a[j] = rp[60 * j];
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) w[jj] = t1[jj * 64 + l];
        }
        DFX_WAVE_SYNC();
        if (active && l < 60) {
            float2 o[8];
            dfx_bfly8<SG>(a, o);
            float2 *wp = x + ((l << 3) + l);
            wp[0] = o[0];
#pragma unroll
            for (int j = 1; j < 8; ++j) wp[j] = dfx_tw_mul<SG>(o[j], w[(j - 1) >> 1], (j - 1) & 1);
        }
        DFX_WAVE_SYNC();
    }
    {   // ---- pass 2: 80 butterflies of radix 6 (b = 8 p + q; the second round: b = 64 + l for l < 16), slot b + (b >> 3) + 90 j -> slot q + 54 p + 9 j
        int l = lane;
        DFX_OPAQUE(l);
        float2 a[2][6];
        f32x4 w[2][3];
        const int l1 = l & 15, p0 = l >> 3, p1 = l1 >> 3;
        {
            const float2 *r0 = x + l + p0, *r1 = x + 72 + l1 + p1;
#pragma unroll
            for (int j = 0; j < 6; ++j) a[0][j] = r0[90 * j];
#pragma unroll
            for (int j = 0; j < 6; ++j) a[1][j] = r1[90 * j];
#pragma unroll
            for (int jj = 0; jj < 3; ++jj) w[0][jj] = t2[jj * 10 + p0], w[1][jj] = t2[jj * 10 + 8 + p1];
        }
        DFX_WAVE_SYNC();
        if (active) {
            float2 o[6];
            dfx_bfly6<SG>(a[0], o);
            float2 *wp = x + (l & 7) + DFX_MUL24(p0, 54);
            wp[0] = o[0];
#pragma unroll
            for (int j = 1; j < 6; ++j) wp[9 * j] = dfx_tw_mul<SG>(o[j], w[0][(j - 1) >> 1], (j - 1) & 1);
            if (l < 16) {
                dfx_bfly6<SG>(a[1], o);
                wp = x + 432 + (l & 7) + DFX_MUL24(p1, 54);
                wp[0] = o[0];
#pragma unroll
                for (int j = 1; j < 6; ++j) wp[9 * j] = dfx_tw_mul<SG>(o[j], w[1][(j - 1) >> 1], (j - 1) & 1);
            }
        }
        DFX_WAVE_SYNC();
    }
    {   // ---- pass 3: 48 butterflies of radix 10, slot q + (q >> 3) + 54 j (padded) -> element q + 48 j (natural layout), no twiddles
        int l = lane;
        DFX_OPAQUE(l);
        float2 a[10];
        {
            const int q = l < 48 ? l : 47;
            const float2 *rp = x + q + (q >> 3);
#pragma unroll
            for (int j = 0; j < 10; ++j) a[j] = rp[54 * j];
        }
        DFX_WAVE_SYNC();
        if (active && l < 48) {
            float2 o[10];
            dfx_bfly10<SG>(a, o);
            float2 *wp = x + l;
#pragma unroll
            for (int j = 0; j < 10; ++j) wp[48 * j] = o[j];
        }
        DFX_WAVE_SYNC();
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// The 480-point transform on the matrix pipe (round 5).  The counters of the VALU form (profiles/r05_pmc_side_kernels_start_of_round.txt)
// show the STFT kernels co-bound by the LDS pipe (82 % busy, half of it bank conflicts of the strided autosort passes) and the VALU
// (66 %: ~480 of a frame's ~830 vector instructions are the three passes, most of them index arithmetic and moves, not flops).  Here the
// transform of ONE frame (one wave, as before: a drop-in for dfx_fft480_ip between the same LDS layouts) is two chained matrix products
// with 16 columns, 480 = 16 x 30, input n = 30 n1 + n2, output k = k1 + 16 k2:
//     Y[n2][k1]  = sum_n1 z[30 n1 + n2] W16^(n1 k1)          A = the frame's data (rows n2, k = (re | im, n1)), B = the 16-point matrix
//     Y'[n2][k1] = Y[n2][k1] W480^(n2 k1)                    eight complex products per lane, the factors lane constants in registers
//     X[k2][k1]  = sum_n2 W30^(n2 k2) Y'[n2][k1]             A = the 30-point matrix (rows k2), B = Y' exactly as the first product's D
//                                                             fragments leave it (lane (k1, q): n2 = 16 (i >> 2) + 4 q + (i & 3))
// 12 + 24 matrix ops of the fp16-split kind (hi hi + hi lo + lo hi, fp32 accumulate) per frame instead of 480 vector instructions and 77
// LDS operations.  Block floating point: the frame is scaled by the power of two that puts its largest component just below 2^14 before
// the first split (any finite input keeps ~22 bits relative to its own peak; no range guard, no fallback), the scales of the constant
// matrices (2^13) and of the twiddle factors (2^-17) are fixed, the result is unscaled on its way back to LDS.  Measured against the
// double-precision transform: 1.5e-7 of the frame's peak (the radix passes in fp32: 3e-8..1e-7) — tools/dev/dft_mfma_check.py.
// Tables (dfx_state::d_mfft, one set per direction): 4 fragments of the 16-point matrix [Br hi, Br lo, Bi hi, Bi lo][64] (registers), 8 of the
// 30-point matrix [Wr | Wi][rows 0..15 | 16..29][hi, lo][64] (staged in LDS by the kernel), 8 complex twiddle factors per lane (registers).
// ---------------------------------------------------------------------------------------------------------------------
#define DFX_MFFT_FRAG1 4
#define DFX_MFFT_FRAG3 8
#define DFX_MFFT_TABLE_BYTES ((size_t)(DFX_MFFT_FRAG1 + DFX_MFFT_FRAG3) * 64 * 16 + (size_t)64 * 8 * 8)   /* per direction */
struct DfxMfftRegs {
    dfx_h8 b1[DFX_MFFT_FRAG1];
    float2 tw[8];
};
static __device__ __forceinline__ void dfx_mfft_load(DfxMfftRegs &R, const unsigned char *table, int lane) {
    const dfx_h8 *fr = reinterpret_cast<const dfx_h8 *>(table);
#pragma unroll
    for (int i = 0; i < DFX_MFFT_FRAG1; ++i) R.b1[i] = fr[i * 64 + lane];
    const float2 *tw = reinterpret_cast<const float2 *>(table + (size_t)(DFX_MFFT_FRAG1 + DFX_MFFT_FRAG3) * 64 * 16);
#pragma unroll
    for (int i = 0; i < 8; ++i) R.tw[i] = tw[lane * 8 + i];
}
// x: the frame's 480 complex values in LDS (natural order), transformed in place; a3: the 30-point fragments in LDS [8][64]
static __device__ __forceinline__ void dfx_fft480_mfma(float2 *x, const DfxMfftRegs &R, const dfx_h8 *a3, int lane, bool active) {
    int l = lane;
    DFX_OPAQUE(l);   // (addresses recomputed per frame instead of living in registers across the frame loop, as in dfx_fft480_ip)
    const int jl = l & 15, q = l >> 4;
    float av[2][8];
    if (active) {
        // A operand of the first product: row n2 = 16 mt + jl (rows 30, 31 re-read row 29: their results meet zero weights), k-slot 8 q + i =
        // (re | im)(z[30 n1 + n2]), n1 = 8 (q & 1) + i, the imaginary parts in the upper half of k
        const float *xf = reinterpret_cast<const float *>(x);
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
            const int n2 = 16 * mt + jl < 30 ? 16 * mt + jl : 29;
            const float *p = xf + 2 * n2 + (q >> 1) + 480 * (q & 1);
#pragma unroll
            for (int i = 0; i < 8; ++i) av[mt][i] = p[60 * i];
        }
    }
    DFX_WAVE_SYNC();   // every read of the frame precedes the first write of the result (in place)
    if (active) {
        float m = 0.f;
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int i = 0; i < 8; ++i) m = fmaxf(m, fabsf(av[mt][i]));
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
        int e = 0;
        if (m > 0.f && m < 3.0e38f) {
            int ex;
            (void)frexpf(m, &ex);   // m = f * 2^ex, f in [0.5, 1)
            e = 14 - ex;
            e = e > 100 ? 100 : (e < -100 ? -100 : e);
        }
        const float sc = ldexpf(1.f, e), us = ldexpf(1.f, -e - 9);
        dfx_h8 ah[2], al[2];
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
#pragma unroll
            for (int i = 0; i < 8; ++i) av[mt][i] *= sc;
            dfx_split8(av[mt], ah[mt], al[mt]);
        }
        // Y (scaled 2^(e + 13)): lane (k1 = jl, q) holds rows n2 = 16 mt + 4 q + r of (Yr | Yi)
        f32x4 y[2][2];
        // (term-major over the four accumulators: a dependent op is four ops behind its predecessor)
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int c = 0; c < 2; ++c) y[mt][c] = dfx_mfma_16x16x32_f16(al[mt], R.b1[2 * c], f32x4{0.f, 0.f, 0.f, 0.f});
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int c = 0; c < 2; ++c) y[mt][c] = dfx_mfma_16x16x32_f16(ah[mt], R.b1[2 * c + 1], y[mt][c]);
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int c = 0; c < 2; ++c) y[mt][c] = dfx_mfma_16x16x32_f16(ah[mt], R.b1[2 * c], y[mt][c]);
        DFX_MFMA_GUARD();
        // twiddle: element 4 mt + r of (yr8 | yi8) = Y'[n2 = 16 mt + 4 q + r][k1] scaled 2^(e - 4): the k order of the second product's B operand
        float yr8[8], yi8[8];
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float2 w = R.tw[4 * mt + r];
                const float a = y[mt][0][r], b = y[mt][1][r];
                yr8[4 * mt + r] = a * w.x - b * w.y;
                yi8[4 * mt + r] = a * w.y + b * w.x;
            }
        dfx_h8 rh, rl, ih, il;
        dfx_split8(yr8, rh, rl);
        dfx_split8(yi8, ih, il);
        // X[k2][k1] (scaled 2^(e + 9)): Xr = Wr Yr' - Wi Yi', Xi = Wi Yr' + Wr Yi'; fragments [Wr | Wi][tile][hi, lo]
        const dfx_h8 *fl = a3 + l;
#pragma unroll
        for (int t2 = 0; t2 < 2; ++t2) {
            const dfx_h8 wrh = fl[((0 * 2 + t2) * 2 + 0) * 64], wrl = fl[((0 * 2 + t2) * 2 + 1) * 64];
            const dfx_h8 wih = fl[((1 * 2 + t2) * 2 + 0) * 64], wil = fl[((1 * 2 + t2) * 2 + 1) * 64];
            // four independent chains of three, term-major: a dependent op is three ops behind its predecessor
            f32x4 pa = dfx_mfma_16x16x32_f16(wrl, rh, f32x4{0.f, 0.f, 0.f, 0.f});   // Wr Yr'
            f32x4 pb = dfx_mfma_16x16x32_f16(wil, ih, f32x4{0.f, 0.f, 0.f, 0.f});   // Wi Yi'
            f32x4 pc = dfx_mfma_16x16x32_f16(wil, rh, f32x4{0.f, 0.f, 0.f, 0.f});   // Wi Yr'
            f32x4 pd = dfx_mfma_16x16x32_f16(wrl, ih, f32x4{0.f, 0.f, 0.f, 0.f});   // Wr Yi'
            pa = dfx_mfma_16x16x32_f16(wrh, rl, pa);
            pb = dfx_mfma_16x16x32_f16(wih, il, pb);
            pc = dfx_mfma_16x16x32_f16(wih, rl, pc);
            pd = dfx_mfma_16x16x32_f16(wrh, il, pd);
            pa = dfx_mfma_16x16x32_f16(wrh, rh, pa);
            pb = dfx_mfma_16x16x32_f16(wih, ih, pb);
            pc = dfx_mfma_16x16x32_f16(wih, rh, pc);
            pd = dfx_mfma_16x16x32_f16(wrh, ih, pd);
        DFX_MFMA_GUARD();
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int k2 = 16 * t2 + 4 * q + r;
                if (k2 < 30) x[jl + 16 * k2] = make_float2((pa[r] - pb[r]) * us, (pc[r] + pd[r]) * us);
            }
        }
    }
    DFX_WAVE_SYNC();
}

// Complex FFT of length pl.M by one 64-lane team.  Data starts in `a`; returns the buffer that holds the result.
// A team is exactly one wave, so the passes only need a wave-level barrier (DFX_WAVE_SYNC): the 8 teams of a workgroup run
// their FFTs independently instead of meeting at a workgroup barrier after every pass.
template <int SG>
static __device__ __forceinline__ float2 *dfx_fft_team(float2 *a, float2 *b, const float2 *tw, const DfxFftPlan &pl,
                                                       int lane, bool active) {
    // the 48 kHz / 20 ms configuration of every shipped model: N = 960, M = 480 = 4*4*2*3*5 (make_plan's order)
    if (pl.M == 480 && pl.nstage == 5 && pl.radix[0] == 4 && pl.radix[1] == 4 && pl.radix[2] == 2 && pl.radix[3] == 3 && pl.radix[4] == 5) {
        if (active) dfx_fft_pass_c<4, SG, 480, 480, 1>(a, b, tw, lane);
        DFX_WAVE_SYNC();
        if (active) dfx_fft_pass_c<4, SG, 480, 120, 4>(b, a, tw, lane);
        DFX_WAVE_SYNC();
        if (active) dfx_fft_pass_c<2, SG, 480, 30, 16>(a, b, tw, lane);
        DFX_WAVE_SYNC();
        if (active) dfx_fft_pass_c<3, SG, 480, 15, 32>(b, a, tw, lane);
        DFX_WAVE_SYNC();
        if (active) dfx_fft_pass_c<5, SG, 480, 5, 96>(a, b, tw, lane);
        DFX_WAVE_SYNC();
        return b;
    }
    int ncur = pl.M, s = 1;
    float2 *x = a, *y = b;
    for (int st = 0; st < pl.nstage; ++st) {
        const int r = pl.radix[st];
        if (active) {
            if (r == 4) dfx_fft_pass<4, SG>(x, y, tw, pl.M, pl.N, ncur, s, lane);
            else if (r == 2) dfx_fft_pass<2, SG>(x, y, tw, pl.M, pl.N, ncur, s, lane);
            else if (r == 3) dfx_fft_pass<3, SG>(x, y, tw, pl.M, pl.N, ncur, s, lane);
            else dfx_fft_pass<5, SG>(x, y, tw, pl.M, pl.N, ncur, s, lane);
        }
        DFX_WAVE_SYNC();
        ncur /= r;
        s *= r;
        float2 *t = x;
        x = y;
        y = t;
    }
    return x;
}

// 16-bit PCM at the boundary (dfx_enhance_pcm16; df/io.py:48,79-80): torchaudio.load's normalisation x / 32768 on the way in, save_audio's
// (audio * (1 << 15)).to(int16) on the way out — truncation toward zero, NaN -> 0, out-of-range values wrap like ATen's
// float -> int64 -> int16 chain.  The same expressions as dfx_k_pcm16_to_f32 / dfx_k_f32_to_pcm16 (dfx_io.hip): same bits.
static __device__ __forceinline__ float dfx_pcm16_in(int16_t v) { return (float)v * (1.0f / 32768.0f); }
static __device__ __forceinline__ int16_t dfx_pcm16_out(float x) {
    float v = x * 32768.0f;
    v = v != v ? 0.f : fminf(fmaxf(v, -9.0e18f), 9.0e18f);
    return (int16_t)(uint16_t)(uint64_t)(int64_t)v;
}
// four consecutive output samples n .. n + 3 of a row (those inside [0, out_len)) as f32 or as 16-bit PCM
template <bool I16>
static __device__ __forceinline__ void dfx_store_out4(float *out_f32, int64_t row_off, int64_t n, int64_t out_len, const f32x4 v) {
    if constexpr (I16) {
        int16_t *o = reinterpret_cast<int16_t *>(out_f32) + row_off + n;
        if (n >= 0 && n + 3 < out_len && (reinterpret_cast<uintptr_t>(o) & 7) == 0) {
            const uint32_t lo = (uint32_t)(uint16_t)dfx_pcm16_out(v[0]) | ((uint32_t)(uint16_t)dfx_pcm16_out(v[1]) << 16);
            const uint32_t hi = (uint32_t)(uint16_t)dfx_pcm16_out(v[2]) | ((uint32_t)(uint16_t)dfx_pcm16_out(v[3]) << 16);
            *reinterpret_cast<uint2 *>(o) = make_uint2(lo, hi);
        } else {
#pragma unroll
            for (int e = 0; e < 4; ++e)
                if (n + e >= 0 && n + e < out_len) o[e] = dfx_pcm16_out(v[e]);
        }
    } else {
        float *o = out_f32 + row_off + n;
        if (n >= 0 && n + 3 < out_len && (reinterpret_cast<uintptr_t>(o) & 15) == 0) {
            *reinterpret_cast<f32x4 *>(o) = v;
        } else {
#pragma unroll
            for (int e = 0; e < 4; ++e)
                if (n + e >= 0 && n + e < out_len) o[e] = v[e];
        }
    }
}

struct DfxAnaArgs {
    const float *x;       // [B, x_stride] (I16 instances: int16_t samples behind the same pointer, strides in samples)
    const float *mem_in;  // [B, N-hop] or null
    float2 *spec;         // [B, Tf, F]
    float *erb_db;        // [B, Tf, nb] or null: 10*log10(band energy + 1e-10)
    const float *window;  // [N]
    const float2 *tw;     // [N]
    const int *band_start;  // [nb+1]
    const float *band_invw; // [nb]
    const int *seg_tab;     // [3*64 + nb + 1] band sums on all 64 lanes (dfx_bands_create): start bin, bins, 1/width (float bits) of each
                            // segment, then the first segment of every band; used when nseg > 0
    int nseg;
    int segcap = 0, segparts = 0;   // longest segment / most segments of a band when the fixed-trip band sums may be used (dfx_launch_analysis), else 0
    int64_t B, Tf, x_stride;
    int64_t x_len;        // samples that exist per row; positions >= x_len read as 0 (enhance()'s F.pad(audio, (0, n_fft)) without a copy)
    int64_t spec_stride;  // row stride of spec in complex elements (>= F; the engine pads odd F to even so rows are 16-byte aligned)
    int hop, nb;
    float wnorm;
    DfxFftPlan plan;
    const unsigned char *mfft = nullptr;   // MF instances: the forward tables of dfx_fft480_mfma
};

// STFT analysis: frame (b,t) = rfft_N( window * stream[(t+1)*hop-N : (t+1)*hop] ) * wnorm   (lib.rs:356-394),
// stream = [mem_in (N-hop zeros after a reset) ; x].  One wave per frame, 8 frames per workgroup pass, grid-stride.
// Optional fused ERB band energies in dB (lib.rs:206-212 without the norm; transforms.rs:236-253).
// IP: the 480-point plan, transformed in place (one LDS buffer per frame; a kernel of its own so that the generic plan's run-time loops
// do not cost it registers: three workgroups = six waves per SIMD have to fit)
// MF: the transform on the matrix pipe (dfx_fft480_mfma; its 30-point fragments, 8 KB, sit behind the window table; four waves per SIMD)
template <bool IP, bool I16 = false, bool MF = false>
__global__ void __launch_bounds__(DFX_DSP_THREADS, IP ? (MF ? 4 : 6) : 4) dfx_k_analysis(DfxAnaArgs A) {
    static_assert(!MF || IP, "the matrix-pipe transform replaces the in-place 480-point plan");
    DFX_DYN_SMEM(unsigned char, smem);
    const int N = A.plan.N, M = A.plan.M, F = M + 1;
    float2 *tw = reinterpret_cast<float2 *>(smem);                       // [N]
    float *win = reinterpret_cast<float *>(smem + (size_t)N * 8);        // [N]
    dfx_h8 *a3s = reinterpret_cast<dfx_h8 *>(smem + (size_t)N * 12);    // [8][64] (MF)
    const size_t team_off = (size_t)N * 12 + (MF ? (size_t)DFX_MFFT_FRAG3 * 64 * 16 : 0);
    const size_t buf_elems = IP ? (size_t)DFX_FFT480_BUF : (size_t)(M + 2);   // M+1 used, padded to keep 16-byte carve (in place: room for the transform's padded layout)
    // (the team = wave index is uniform across the wave: frame index, clip / frame split — a 64-bit division — and the row bases stay scalar)
    const int team = dfx_wave_uniform(threadIdx.x / DFX_DSP_TEAM), lane = threadIdx.x % DFX_DSP_TEAM;
    // the 480-point plan transforms in place: one buffer per frame (the host sizes the dynamic LDS accordingly, dfx_dsp.hip)
    constexpr bool ip = IP;
    float2 *bufA = reinterpret_cast<float2 *>(smem + team_off) + (size_t)team * (ip ? 1 : 2) * buf_elems;
    float2 *bufB = bufA + buf_elems;   // (not ip only)
    // band edges and 1/width of the ERB feature, behind the frames
    int *bstart = reinterpret_cast<int *>(smem + team_off + (size_t)DFX_DSP_TEAMS * (ip ? 1 : 2) * buf_elems * 8);   // [nb + 1]
    float *binvw = reinterpret_cast<float *>(bstart + A.nb + 1);                                                       // [nb]
    int *segs = reinterpret_cast<int *>(binvw + A.nb);                              // [3*64 + nb + 1]  (A.nseg > 0)
    float *part = reinterpret_cast<float *>(segs + 3 * DFX_DSP_TEAM + A.nb + 1) + team * DFX_DSP_TEAM;   // [64] per frame
    if (A.erb_db) {
        for (int i = threadIdx.x; i <= A.nb; i += DFX_DSP_THREADS) {
            bstart[i] = A.band_start[i];
            if (i < A.nb) binvw[i] = A.band_invw[i];
        }
        if (A.nseg > 0)
            for (int i = threadIdx.x; i < 3 * DFX_DSP_TEAM + A.nb + 1; i += DFX_DSP_THREADS) segs[i] = A.seg_tab[i];
    }
    // twiddles + window: every load of a pass is issued before the first LDS store (a load -> store loop waits out one memory
    // latency per iteration; the compiler does not batch across a runtime trip count)
    // T480 (the 480-point plan on the vector pipe): the table area holds the per-pass tables of dfx_fft480_t and, behind them, the post-pass
    // factors tp[k] = exp(-2 pi i k / N) * wnorm / 2 for k <= M / 2 (6.5 of the 7.5 KB of the plain table)
    constexpr bool T480 = IP && !MF;
    f32x4 *t1 = reinterpret_cast<f32x4 *>(smem), *t2 = t1 + DFX_TW480_T1;
    float2 *tp = reinterpret_cast<float2 *>(smem + DFX_TW480_BYTES);   // [M / 2 + 1] (T480)
    const float hnorm = 0.5f * A.wnorm;
    for (int i0 = threadIdx.x; i0 < N; i0 += 4 * DFX_DSP_THREADS) {
        float2 tv[4];
        float wv[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int i = i0 + u * DFX_DSP_THREADS;
            tv[u] = i < N && (!T480 || i <= 240) ? A.tw[i] : make_float2(0.f, 0.f);
            wv[u] = i < N ? A.window[i] : 0.f;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int i = i0 + u * DFX_DSP_THREADS;
            if (i < N) {
                if constexpr (T480) {
                    if (i <= 240) tp[i] = make_float2(tv[u].x * hnorm, tv[u].y * hnorm);
                } else {
                    tw[i] = tv[u];
                }
                win[i] = wv[u];
            }
        }
    }
    if constexpr (T480) dfx_tw480_fill(t1, t2, A.tw, (int)threadIdx.x, DFX_DSP_THREADS);
    DfxMfftRegs mfr;
    if constexpr (MF) {
        for (int i = threadIdx.x; i < DFX_MFFT_FRAG3 * 64; i += DFX_DSP_THREADS) a3s[i] = reinterpret_cast<const dfx_h8 *>(A.mfft)[DFX_MFFT_FRAG1 * 64 + i];
        dfx_mfft_load(mfr, A.mfft, lane);
    }
    __syncthreads();
    const int64_t nframes = A.B * A.Tf;
    const int ML = N - A.hop;
    for (int64_t base = (int64_t)blockIdx.x * DFX_DSP_TEAMS; base < nframes; base += (int64_t)gridDim.x * DFX_DSP_TEAMS) {
        const int64_t fr = base + team;
        const bool active = fr < nframes;
        const int64_t b = active ? fr / A.Tf : 0, t = active ? fr - b * A.Tf : 0;
        if (active) {
            const float *xb = A.x + b * A.x_stride;
            const int16_t *xs = reinterpret_cast<const int16_t *>(A.x) + b * A.x_stride;   // (I16)
            const int64_t pos0 = t * A.hop - ML;
            const float *xf = xb + pos0;
            int ll = lane;
            DFX_OPAQUE(ll);   // (addresses recomputed per frame instead of living in registers across the loop: see dfx_fft480_ip)
            DFX_ASSUME(ll >= 0 && ll < DFX_DSP_TEAM);
            if (pos0 >= 0 && pos0 + N <= A.x_len && (I16 ? (reinterpret_cast<uintptr_t>(xs + pos0) & 3) == 0 : (reinterpret_cast<uintptr_t>(xf) & 7) == 0)) {
                // interior frame (wave-uniform test; all but the first of a clip and the ones reaching into the implicit zero padding):
                // 8-byte loads, 8 per lane in flight before the first LDS store (M = 480: one pass) — a load -> store loop would
                // wait out one memory latency per iteration.  (Requesting the NEXT frame before this one's FFT, as the synthesis
                // kernel does, was measured slower here: 16 more live registers cost the second resident workgroup: 0.76 -> 1.14 ms.)
                const float2 *xf2 = reinterpret_cast<const float2 *>(xf);
                for (int k0 = ll; k0 < M; k0 += 8 * DFX_DSP_TEAM) {
                    float2 v[8], wv[8];   // (the window too: an LDS read between LDS stores is a round trip of its own, see dfx_fft_pass_ip)
#pragma unroll
                    for (int u = 0; u < 8; ++u) {
                        const int k = k0 + u * DFX_DSP_TEAM;
                        if constexpr (I16) {   // two samples per 4-byte load
                            const uint32_t pv = reinterpret_cast<const uint32_t *>(xs + pos0)[k < M ? k : k0];
                            v[u] = make_float2(dfx_pcm16_in((int16_t)(pv & 0xffffu)), dfx_pcm16_in((int16_t)(pv >> 16)));
                        } else {
                            v[u] = xf2[k < M ? k : k0];
                        }
                        wv[u] = reinterpret_cast<const float2 *>(win)[k < M ? k : k0];
                    }
#pragma unroll
                    for (int u = 0; u < 8; ++u) {
                        const int k = k0 + u * DFX_DSP_TEAM;
                        if (k < M) bufA[k] = make_float2(v[u].x * wv[u].x, v[u].y * wv[u].y);
                    }
                }
            } else {
                for (int k = ll; k < M; k += DFX_DSP_TEAM) {
                    float v[2];
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        const int i = 2 * k + h;
                        const int64_t pos = pos0 + i;
                        float s = 0.f;
                        if (pos >= 0) s = pos < A.x_len ? (I16 ? dfx_pcm16_in(xs[pos]) : xb[pos]) : 0.f;
                        else if (A.mem_in) s = A.mem_in[b * ML + (ML + pos)];
                        v[h] = s * win[i];
                    }
                    bufA[k] = make_float2(v[0], v[1]);
                }
            }
        }
        DFX_WAVE_SYNC();
        float2 *Z = bufA;
        float *pw = nullptr;    // |X|^2 per bin for the ERB feature (float array)
        int pws = 1;            // stride of pw in floats
        if constexpr (IP) {
            if constexpr (MF) dfx_fft480_mfma(bufA, mfr, a3s, lane, active);
            else dfx_fft480_t<-1>(bufA, t1, t2, lane, active);
            // real post-pass on the pairs (k, M-k): both bins of a pair need Z[k] and Z[M-k] and nothing else, so a lane that owns the pair
            // can put |X|^2 of the two bins back into the .x halves of the two slots it has just read (pw stride 2 floats; slot M takes
            // the Nyquist bin)
            pw = reinterpret_cast<float *>(bufA);
            pws = 2;
            if (active) {
                float2 *out = A.spec + (b * A.Tf + t) * A.spec_stride;
                if (lane == 0 && A.spec_stride > F) out[F] = make_float2(0.f, 0.f);  // the pad bin of an aligned row
                // E = (Z[k] + conj(Z[M-k]))/2 ; O = (Z[k] - conj(Z[M-k]))/(2i) ; X[k] = (E + exp(-2*pi*i*k/N) * O) * wnorm.  The two bins of a pair share
                // E and O up to signs and exp(-2 pi i (M - k) / N) = -conj(exp(-2 pi i k / N)): with tt = (Z[k].y + Z[M-k].y, Z[M-k].x - Z[k].x) * tp[k],
                // X[k] = ((er, ei) * wnorm / 2) + tt and X[M-k] = ((er, -ei) * wnorm / 2) + (-tt.x, tt.y) — 12 operations per pair where the two
                // separate evaluations took 28 (the halves are exact: only the scaled factor and the fused multiply-adds round differently)
                // (every LDS read of the pass before its first LDS store, as in the transform's passes)
                constexpr int MI = 480, NPR = (MI / 2 + DFX_DSP_TEAM) / DFX_DSP_TEAM;
                float2 za[NPR], zb[NPR], ta[NPR];
                int lp = lane;
                DFX_OPAQUE(lp);
                DFX_ASSUME(lp >= 0 && lp < DFX_DSP_TEAM);
                // pairs k = lane + 64 i: i < 3 always exist (k <= 191), i = 3 for the lanes with k <= MI / 2 (the others read pair 0 and store nothing);
                // only the last pair of lane 48 is its own partner (round 6: was a clamp, a k <= MI / 2 test and a partner test in every iteration)
                const bool last = lp + (NPR - 1) * DFX_DSP_TEAM <= MI / 2;
#pragma unroll
                for (int i = 0; i < NPR; ++i) {
                    const int kk = (i < NPR - 1 || last) ? lp + i * DFX_DSP_TEAM : 0, kc = MI - kk;
                    za[i] = Z[kk], zb[i] = Z[(i == 0 && kk == 0) ? 0 : kc];   // partner bin (k = 0: the Nyquist bin M, both from Z[0])
                    if constexpr (T480) ta[i] = tp[kk];
                    else ta[i] = make_float2(tw[kk].x * hnorm, tw[kk].y * hnorm);
                }
#pragma unroll
                for (int i = 0; i < NPR; ++i) {
                    const int k = lp + i * DFX_DSP_TEAM, kc = MI - k;
                    if (i == NPR - 1 && !last) continue;
                    const float er = za[i].x + zb[i].x, ei = za[i].y - zb[i].y;
                    const float dr = za[i].x - zb[i].x, di = za[i].y + zb[i].y;
                    const float2 tt = dfx_cmul(make_float2(di, -dr), ta[i]);
                    const float2 Xa = make_float2(fmaf(er, hnorm, tt.x), fmaf(ei, hnorm, tt.y));
                    const float2 Xb = make_float2(fmaf(er, hnorm, -tt.x), fmaf(-ei, hnorm, tt.y));
                    const float pa = __fadd_rn(__fmul_rn(Xa.x, Xa.x), __fmul_rn(Xa.y, Xa.y)), pb = __fadd_rn(__fmul_rn(Xb.x, Xb.x), __fmul_rn(Xb.y, Xb.y));
                    if (i < NPR - 1 || kc != k) out[kc] = Xb;
                    out[k] = Xa;
                    if (A.erb_db) {   // (LDS operations of a wave complete in order: where a pair is its own partner the second store stays)
                        pw[2 * kc] = pb;
                        pw[2 * k] = pa;
                    }
                }
            }
        } else {
        Z = dfx_fft_team<-1>(bufA, bufB, tw, A.plan, lane, active);
        float2 *other = (Z == bufA) ? bufB : bufA;
        pw = reinterpret_cast<float *>(other);
        if (active) {
            float2 *out = A.spec + (b * A.Tf + t) * A.spec_stride;
            if (lane == 0 && A.spec_stride > F) out[F] = make_float2(0.f, 0.f);  // the pad bin of an aligned row
            for (int k = lane; k <= M; k += DFX_DSP_TEAM) {
                const float2 zk = Z[k == M ? 0 : k];
                const float2 zc = Z[k == 0 ? 0 : M - k];
                // E = (Z[k] + conj(Z[M-k]))/2 ; O = (Z[k] - conj(Z[M-k]))/(2i) ; X[k] = E + exp(-2*pi*i*k/N) * O
                const float er = 0.5f * (zk.x + zc.x), ei = 0.5f * (zk.y - zc.y);
                const float dr = 0.5f * (zk.x - zc.x), di = 0.5f * (zk.y + zc.y);
                const float2 tt = dfx_cmul(make_float2(di, -dr), tw[k]);
                const float2 X = make_float2((er + tt.x) * A.wnorm, (ei + tt.y) * A.wnorm);
                out[k] = X;
                if (A.erb_db) pw[k] = __fadd_rn(__fmul_rn(X.x, X.x), __fmul_rn(X.y, X.y));
            }
        }
        }
        if (A.erb_db) {
            DFX_WAVE_SYNC();
            if (A.nseg > 0 && A.segcap > 0) {
                // The same segment sums with FIXED trip counts (round 6): every lane reads segcap values from `segment start + constant` and adds
                // the ones inside its segment (the rest are in-range slots of the frame buffer), the 1/width factor once per segment; the band's lane
                // then adds at most segparts segment sums the same way.  No per-element index clamps, predicated reads or loop-carried
                // addresses: ~90 vector instructions per frame where the loops below take ~230 (a quarter of the kernel, which is VALU-bound).
                int le = lane;
                DFX_OPAQUE(le);
                DFX_ASSUME(le >= 0 && le < DFX_DSP_TEAM);
                const int n = segs[DFX_DSP_TEAM + le];   // (0 for lanes beyond the last segment: the table is zero-filled)
                const float kk = __int_as_float(segs[2 * DFX_DSP_TEAM + le]);
                const float *pp = pw + pws * segs[le];
                float acc = 0.f;
                for (int j0 = 0; j0 < A.segcap; j0 += 6) {
                    float pv[6];
#pragma unroll
                    for (int u = 0; u < 6; ++u) pv[u] = pp[pws * u];
#pragma unroll
                    for (int u = 0; u < 6; ++u) acc = __fadd_rn(acc, j0 + u < n ? pv[u] : 0.f);
                    pp += 6 * pws;
                }
                if (active) part[le] = __fmul_rn(acc, kk);
                DFX_WAVE_SYNC();
                {
                    const int *bseg = segs + 3 * DFX_DSP_TEAM;
                    const int lb = le < A.nb ? le : 0;
                    const int g0 = bseg[lb], cnt = bseg[lb + 1] - g0;
                    const float *qq = part + g0;
                    float tot = 0.f;
                    for (int u0 = 0; u0 < A.segparts; u0 += 4) {
                        float pv[4];
#pragma unroll
                        for (int u = 0; u < 4; ++u) pv[u] = qq[u];
#pragma unroll
                        for (int u = 0; u < 4; ++u) tot = __fadd_rn(tot, u0 + u < cnt ? pv[u] : 0.f);
                        qq += 4;
                    }
                    if (active && le < A.nb) A.erb_db[(b * A.Tf + t) * A.nb + le] = log10f(tot + 1e-10f) * 10.f;
                }
            } else if (A.nseg > 0) {
                // compute_band_corr (lib.rs:280-295): acc += |X|^2 * (1/width) over the bins of a band.  The bands are cut into at most 64
                // segments of near-equal length (11 bins at 48 kHz, where the widest band has 67): every lane sums one segment, bins in
                // ascending order, then the lane of a band adds its segments in ascending order.  (One lane per band, as below, makes the
                // widest band the length of the whole stage — a third of this kernel's instructions with half of the lanes idle.)
                int le = lane;
                DFX_OPAQUE(le);
                DFX_ASSUME(le >= 0 && le < DFX_DSP_TEAM);
                const bool mine = active && le < A.nseg;
                const int s0 = segs[mine ? le : 0], n = mine ? segs[DFX_DSP_TEAM + le] : 0;
                const float kk = __int_as_float(segs[2 * DFX_DSP_TEAM + (mine ? le : 0)]);
                float acc = 0.f;
                for (int j0 = 0; j0 < n; j0 += 6) {
                    float pv[6];
#pragma unroll
                    for (int u = 0; u < 6; ++u) pv[u] = pw[pws * (s0 + (j0 + u < n ? j0 + u : 0))];
#pragma unroll
                    for (int u = 0; u < 6; ++u)
                        if (j0 + u < n) acc = __fadd_rn(acc, __fmul_rn(pv[u], kk));
                }
                if (mine) part[le] = acc;
                DFX_WAVE_SYNC();
                if (active && le < A.nb) {
                    const int *bseg = segs + 3 * DFX_DSP_TEAM;
                    const int g0 = bseg[le], g1 = bseg[le + 1];
                    float tot = 0.f;
                    for (int g = g0; g < g1; g += 8) {
                        float pv[8];
#pragma unroll
                        for (int u = 0; u < 8; ++u) pv[u] = part[g + u < g1 ? g + u : g0];
#pragma unroll
                        for (int u = 0; u < 8; ++u)
                            if (g + u < g1) tot = __fadd_rn(tot, pv[u]);
                    }
                    A.erb_db[(b * A.Tf + t) * A.nb + le] = log10f(tot + 1e-10f) * 10.f;
                }
            } else if (active && lane < A.nb) {
                // compute_band_corr (lib.rs:280-295): acc += |X|^2 * (1/width), bins in ascending order.  Eight bins are read before they are
                // added (same order, same sum): with one read per trip the widest band — 67 bins at 48 kHz — was 67 dependent LDS round
                // trips, more than the five passes of the transform together
                int le = lane;
                DFX_OPAQUE(le);
                DFX_ASSUME(le >= 0 && le < DFX_DSP_TEAM);
                const int s0 = bstart[le], s1 = bstart[le + 1];
                const float kk = binvw[le];
                float acc = 0.f;
                for (int j0 = s0; j0 < s1; j0 += 8) {
                    float pv[8];
#pragma unroll
                    for (int u = 0; u < 8; ++u) pv[u] = pw[pws * (j0 + u < s1 ? j0 + u : s0)];
#pragma unroll
                    for (int u = 0; u < 8; ++u)
                        if (j0 + u < s1) acc = __fadd_rn(acc, __fmul_rn(pv[u], kk));
                }
                A.erb_db[(b * A.Tf + t) * A.nb + le] = log10f(acc + 1e-10f) * 10.f;
            }
            for (int e = DFX_DSP_TEAM + lane; A.nseg <= 0 && active && e < A.nb; e += DFX_DSP_TEAM) {  // nb > 64 (rare)
                const int s0 = bstart[e], s1 = bstart[e + 1];
                const float kk = binvw[e];
                float acc = 0.f;
                for (int j = s0; j < s1; ++j) acc = __fadd_rn(acc, __fmul_rn(pw[pws * j], kk));
                A.erb_db[(b * A.Tf + t) * A.nb + e] = log10f(acc + 1e-10f) * 10.f;
            }
        }
        DFX_WAVE_SYNC();
    }
}

// analysis memory after the last frame = the last N-hop samples of [mem_in ; x[:, :Tf*hop]]  (lib.rs:379-384)
__global__ void dfx_k_analysis_mem_out(const float *x, const float *mem_in, float *mem_out, int64_t B, int64_t Tf,
                                       int64_t x_stride, int hop, int ML) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B * ML) return;
    const int64_t b = i / ML;
    const int j = (int)(i - b * ML);
    const int64_t pos = Tf * hop - ML + j;
    float v = 0.f;
    if (pos >= 0) v = x[b * x_stride + pos];
    else if (mem_in) v = mem_in[b * ML + (ML + pos)];
    mem_out[i] = v;
}

struct DfxSynArgs {
    const float2 *spec;   // [B, Tf, F]
    const float *mem_in;  // [B, N-hop] or null
    float *mem_out;       // [B, N-hop] or null
    float *out;           // [B, out_stride] (I16 instances: int16_t samples behind the same pointer)
    const float *window;
    const float2 *tw;
    int64_t B, Tf, out_stride;
    int64_t spec_stride;        // row stride of spec in complex elements (>= F)
    int64_t f_begin, f_end;     // output frames [f_begin, f_end) are produced by this launch (f_end may include the R-1 memory frames)
    int64_t out_skip, out_len;  // only stream samples [out_skip, out_skip + out_len) are stored, at out[row][n - out_skip]
    int hop, R /* N/hop rounded up: frames overlapping one output hop */, outf /* output frames per chunk */;
    int chunks;           // chunks per row (including the tail chunk that produces mem_out)
    DfxFftPlan plan;
    const unsigned char *mfft = nullptr;   // MF instances: the inverse tables of dfx_fft480_mfma
};

// ISTFT + window + overlap-add (lib.rs:396-427).  A workgroup produces `outf` consecutive output hops of one row from
// DFX_DSP_TEAMS = outf + R - 1 frames (the first R-1 are halo frames recomputed instead of carried through memory).
// Sum order per output sample follows the reference: oldest contribution first, the current frame last.
// IP: the 480-point plan in place, one LDS buffer per frame (see dfx_k_analysis<IP>).
template <bool IP, bool I16 = false, bool MF = false>
__global__ void __launch_bounds__(DFX_DSP_THREADS, IP ? (MF ? 4 : 6) : 4) dfx_k_synthesis(DfxSynArgs A) {
    static_assert(!MF || IP, "the matrix-pipe transform replaces the in-place 480-point plan");
    DFX_DYN_SMEM(unsigned char, smem);
    const int N = A.plan.N, M = A.plan.M;
    float2 *tw = reinterpret_cast<float2 *>(smem);
    float *win = reinterpret_cast<float *>(smem + (size_t)N * 8);
    dfx_h8 *a3s = reinterpret_cast<dfx_h8 *>(smem + (size_t)N * 12);    // [8][64] (MF)
    const size_t team_off = (size_t)N * 12 + (MF ? (size_t)DFX_MFFT_FRAG3 * 64 * 16 : 0);
    DfxMfftRegs mfr;
    if constexpr (MF) {
        for (int i = threadIdx.x; i < DFX_MFFT_FRAG3 * 64; i += DFX_DSP_THREADS) a3s[i] = reinterpret_cast<const dfx_h8 *>(A.mfft)[DFX_MFFT_FRAG1 * 64 + i];
        dfx_mfft_load(mfr, A.mfft, (int)(threadIdx.x % DFX_DSP_TEAM));
    }
    const size_t buf_elems = IP ? (size_t)DFX_FFT480_BUF : (size_t)(M + 2);
    constexpr int NBUF = IP ? 1 : 2;   // buffers per frame
    const int team = dfx_wave_uniform(threadIdx.x / DFX_DSP_TEAM), lane = threadIdx.x % DFX_DSP_TEAM;   // (wave-uniform: scalar frame / row arithmetic)
    float2 *bufs = reinterpret_cast<float2 *>(smem + team_off);
    float2 *bufA = bufs + (size_t)team * NBUF * buf_elems;
    float2 *bufB = IP ? bufA : bufA + buf_elems;   // the frame is staged here
    // the last 4 bytes of the team area of team 0 would be too fragile for a flag: keep result-buffer parity in a
    // register instead (identical for all teams because the plan is uniform)
    // twiddles + window: every load of a pass is issued before the first LDS store (a load -> store loop waits out one memory
    // latency per iteration; the compiler does not batch across a runtime trip count)
    for (int i0 = threadIdx.x; i0 < N; i0 += 4 * DFX_DSP_THREADS) {
        float2 tv[4];
        float wv[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int i = i0 + u * DFX_DSP_THREADS;
            tv[u] = i < N ? A.tw[i] : make_float2(0.f, 0.f);
            wv[u] = i < N ? A.window[i] : 0.f;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int i = i0 + u * DFX_DSP_THREADS;
            if (i < N) {
                tw[i] = tv[u];
                win[i] = wv[u];
            }
        }
    }
    __syncthreads();
    const int ML = N - A.hop;
    // persistent workgroups: the tables above are staged once, then the workgroup walks (row, chunk) work items grid-stride
    const bool single = M + 1 <= 8 * DFX_DSP_TEAM;
    float2 pre[8];
    bool have_pre = false;
    auto request = [&](const float2 *Y) {
        int lr = lane;
        DFX_OPAQUE(lr);   // (per-frame address / bounds arithmetic instead of registers held across the work-item loop: see dfx_fft480_ip)
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int k = lr + u * DFX_DSP_TEAM;
            pre[u] = Y[k <= M ? k : lr];
        }
    };
    for (int64_t item = blockIdx.x; item < A.B * A.chunks; item += gridDim.x) {
    const int64_t b = item / A.chunks;
    const int chunk = (int)(item - b * A.chunks);
    const int64_t t0 = A.f_begin + (int64_t)chunk * A.outf;  // first output frame of this chunk
    const int64_t t = t0 - (A.R - 1) + team;         // real frame handled by this team
    const bool active = t >= 0 && t < A.Tf;
    if (active) {
        const float2 *Y = A.spec + (b * A.Tf + t) * A.spec_stride;
        if (single) {  // F <= 512: the frame was requested while the previous work item was being transformed (or right now)
            if (!have_pre) request(Y);
            int ls = lane;
            DFX_OPAQUE(ls);
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int k = ls + u * DFX_DSP_TEAM;
                if (k <= M) bufB[k] = pre[u];
            }
        } else {
            for (int k0 = lane; k0 <= M; k0 += 8 * DFX_DSP_TEAM) {  // 8 loads per lane in flight before the first LDS store
                float2 v[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int k = k0 + u * DFX_DSP_TEAM;
                    v[u] = k <= M ? Y[k] : make_float2(0.f, 0.f);
                }
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int k = k0 + u * DFX_DSP_TEAM;
                    if (k <= M) bufB[k] = v[u];
                }
            }
        }
    }
    have_pre = false;
    {   // request this team's frame of the next work item: its HBM latency hides behind the transform and the overlap-add below
        const int64_t nitem = item + gridDim.x;
        if (!IP && single && nitem < A.B * A.chunks) {   // (in place: six waves per SIMD cover the latency; the 16 prefetch registers would spill)
            const int64_t nb = nitem / A.chunks;
            const int64_t nt = A.f_begin + (nitem - nb * A.chunks) * A.outf - (A.R - 1) + team;
            if (nt >= 0 && nt < A.Tf) {
                request(A.spec + (nb * A.Tf + nt) * A.spec_stride);
                have_pre = true;
            }
        }
    }
    DFX_WAVE_SYNC();
    // E' = X[k] + conj(X[M-k]) ; O' = conj(w^k) * (X[k] - conj(X[M-k])) ; Z[k] = E' + i*O'
    auto zbin_w = [&](int k, float2 xk, float2 xm, float2 w) -> float2 {
        if (k == 0) {  // C2R ignores imag(DC) and imag(Nyquist)
            xk.y = 0.f;
            xm.y = 0.f;
        }
        const float er = xk.x + xm.x, ei = xk.y - xm.y;
        w.y = -w.y;
        const float2 o = dfx_cmul(make_float2(xk.x - xm.x, xk.y + xm.y), w);
        return make_float2(er - o.y, ei + o.x);
    };
    auto zbin = [&](int k, float2 xk, float2 xm) -> float2 { return zbin_w(k, xk, xm, tw[k]); };
    float2 *Z = bufA;
    if constexpr (IP) {
        // in place on the pairs (k, M-k): Z[k] and Z[M-k] need X[k] and X[M-k] and nothing else (k = 0 pairs with the Nyquist bin M and
        // only produces Z[0]; k = M/2 is its own partner)
        // (every LDS read of the pass before its first LDS store: see dfx_fft_pass_ip)
        if (active) {
            constexpr int MI = 480, NPR = (MI / 2 + DFX_DSP_TEAM) / DFX_DSP_TEAM;
            float2 xa[NPR], xb[NPR], wa[NPR], wb[NPR];
            int lp = lane;
            DFX_OPAQUE(lp);
#pragma unroll
            for (int i = 0; i < NPR; ++i) {
                const int k = lp + i * DFX_DSP_TEAM, kk = k <= MI / 2 ? k : 0, kc = MI - kk;
                xa[i] = bufA[kk], xb[i] = bufA[kc];
                wa[i] = tw[kk], wb[i] = tw[kc];
            }
#pragma unroll
            for (int i = 0; i < NPR; ++i) {
                const int k = lp + i * DFX_DSP_TEAM, kc = MI - k;
                if (k > MI / 2) continue;
                const float2 za = zbin_w(k, xa[i], xb[i], wa[i]);
                if (k != 0 && kc != k) bufA[kc] = zbin_w(kc, xb[i], xa[i], wb[i]);
                bufA[k] = za;
            }
        }
        DFX_WAVE_SYNC();
        if constexpr (MF) dfx_fft480_mfma(bufA, mfr, a3s, lane, active);
        else dfx_fft480_ip<+1>(bufA, tw, lane, active);
    } else {
        if (active)
            for (int k = lane; k < M; k += DFX_DSP_TEAM) bufA[k] = zbin(k, bufB[k], bufB[M - k]);
        DFX_WAVE_SYNC();
        Z = dfx_fft_team<+1>(bufA, bufB, tw, A.plan, lane, active);
    }
    const bool in_a = (Z == bufA);
    {
        // apply_window_in_place (lib.rs:406): the interleaved (re, im) pairs of z ARE the time samples
        float *xt = reinterpret_cast<float *>(Z);
        if constexpr (IP) {
            // four samples per access, all reads before the first store (a read-multiply-store loop is one LDS round trip per trip: 15)
            constexpr int NQ = 960 / 4, NR4 = (NQ + DFX_DSP_TEAM - 1) / DFX_DSP_TEAM;
            f32x4 *xq = reinterpret_cast<f32x4 *>(xt);
            const f32x4 *wq = reinterpret_cast<const f32x4 *>(win);
            f32x4 xv[NR4], wv[NR4];
            int lw = lane;
            DFX_OPAQUE(lw);
            if (active) {
#pragma unroll
                for (int r = 0; r < NR4; ++r) {
                    const int i = lw + r * DFX_DSP_TEAM, ii = i < NQ ? i : 0;
                    xv[r] = xq[ii], wv[r] = wq[ii];
                }
#pragma unroll
                for (int r = 0; r < NR4; ++r) {
                    const int i = lw + r * DFX_DSP_TEAM;
                    if (i < NQ) xq[i] = xv[r] * wv[r];
                }
            }
        } else if (active)
            for (int i = lane; i < N; i += DFX_DSP_TEAM) xt[i] *= win[i];
    }
    __syncthreads();
    // overlap-add: output frame tf = t0 + j (j < outf), sample i < hop, gets real frames tf-R+1 .. tf
    const int total = A.outf * A.hop;
    const bool quads = (A.hop & 3) == 0 && (N & 3) == 0;   // four consecutive samples share every index and bound below
    if (quads) {
        // the same sums on four samples at a time: one set of index arithmetic, 16-byte LDS reads and (where the row position allows)
        // 16-byte stores per four samples — the scalar form below spent more instructions on its indices than the transform on its data
        const int hq = A.hop >> 2, totq = A.outf * hq;
        const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
        for (int q = threadIdx.x; q < totq; q += DFX_DSP_THREADS) {
            const int j = q / hq, i = (q - j * hq) << 2;
            const int64_t tf = t0 + j;
            if (tf >= A.Tf + (A.mem_out ? A.R - 1 : 0) || tf >= A.f_end) break;
            const int64_t s_glob = tf * A.hop + i;
            f32x4 acc = zero4;
            bool have = false;
            if (A.mem_in && s_glob < ML) {
                const float *mi = A.mem_in + b * ML + s_glob;
                acc = f32x4{mi[0], mi[1], mi[2], mi[3]};
                have = true;
            }
            for (int r = A.R - 1; r >= 1; --r) {  // older frames first
                const int64_t tr = tf - r;
                const int off = r * A.hop + i;
                if (tr >= 0 && tr < A.Tf && off < N) {
                    const int tm = (int)(tr - (t0 - (A.R - 1)));
                    const float *fr = reinterpret_cast<const float *>(bufs + (size_t)tm * NBUF * buf_elems + (in_a ? 0 : buf_elems));
                    const f32x4 fv = *reinterpret_cast<const f32x4 *>(fr + off);
                    acc = have ? acc + fv : fv;
                    have = true;
                }
            }
            f32x4 cur = zero4;
            if (tf < A.Tf) {
                const int tm = (int)(tf - (t0 - (A.R - 1)));
                const float *fr = reinterpret_cast<const float *>(bufs + (size_t)tm * NBUF * buf_elems + (in_a ? 0 : buf_elems));
                cur = *reinterpret_cast<const f32x4 *>(fr + i);
            }
            const f32x4 v = have ? cur + acc : cur;
            if (tf < A.Tf) {
                dfx_store_out4<I16>(A.out, b * A.out_stride, s_glob - A.out_skip, A.out_len, v);
            } else {
                const int64_t mj = s_glob - A.Tf * A.hop;
                if (mj < ML) {
                    float *o = A.mem_out + b * ML + mj;
#pragma unroll
                    for (int e = 0; e < 4; ++e) o[e] = v[e];
                }
            }
        }
    }
    for (int idx = threadIdx.x; !quads && idx < total; idx += DFX_DSP_THREADS) {
        const int j = idx / A.hop, i = idx - j * A.hop;
        const int64_t tf = t0 + j;
        if (tf >= A.Tf + (A.mem_out ? A.R - 1 : 0) || tf >= A.f_end) break;
        const int64_t s_glob = tf * A.hop + i;  // sample index in the row's output stream
        float acc = 0.f;
        bool have = false;
        if (A.mem_in && s_glob < ML) {
            acc = A.mem_in[b * ML + s_glob];
            have = true;
        }
        for (int r = A.R - 1; r >= 1; --r) {  // older frames first
            const int64_t tr = tf - r;
            const int off = r * A.hop + i;
            if (tr >= 0 && tr < A.Tf && off < N) {
                const int tm = (int)(tr - (t0 - (A.R - 1)));
                const float *fr = reinterpret_cast<const float *>(bufs + (size_t)tm * NBUF * buf_elems + (in_a ? 0 : buf_elems));
                acc = have ? acc + fr[off] : fr[off];
                have = true;
            }
        }
        float cur = 0.f;
        if (tf < A.Tf) {
            const int tm = (int)(tf - (t0 - (A.R - 1)));
            const float *fr = reinterpret_cast<const float *>(bufs + (size_t)tm * NBUF * buf_elems + (in_a ? 0 : buf_elems));
            cur = fr[i];
        }
        const float v = have ? cur + acc : cur;
        if (tf < A.Tf) {
            const int64_t n = s_glob - A.out_skip;
            if (n >= 0 && n < A.out_len) {
                if constexpr (I16) reinterpret_cast<int16_t *>(A.out)[b * A.out_stride + n] = dfx_pcm16_out(v);
                else A.out[b * A.out_stride + n] = v;
            }
        } else {
            const int64_t mj = s_glob - A.Tf * A.hop;
            if (mj < ML) A.mem_out[b * ML + mj] = v;
        }
    }
    __syncthreads();  // the team buffers are reused by the next work item
    }
}

// erb() (transforms.rs:236-253 / lib.rs:280-295): one thread per (row, band); rows of F complex bins.
__global__ void dfx_k_erb(const float2 *spec, int64_t rows, int F, int nb, const int *band_start,
                          const float *band_invw, int db, float *out) {
#pragma clang fp contract(off)  // lib.rs:280-295 evaluates |x|^2 * k and the running sum as separate f32 operations
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= rows * nb) return;
    const int64_t r = i / nb;
    const int e = (int)(i - r * nb);
    const float2 *x = spec + r * F;
    const float kk = band_invw[e];
    float acc = 0.f;
    for (int j = band_start[e]; j < band_start[e + 1]; ++j) {
        const float2 v = x[j];
        acc += (v.x * v.x + v.y * v.y) * kk;
    }
    out[i] = db ? log10f(acc + 1e-10f) * 10.f : acc;
}

// erb_inv() (lib.rs:339-348): out[row, f] = gains[row, band(f)]
__global__ void dfx_k_erb_inv(const float *gains, int64_t rows, int F, int nb, const unsigned char *bin2band, float *out) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= rows * F) return;
    const int64_t r = i / F;
    const int f = (int)(i - r * F);
    out[i] = gains[r * nb + bin2band[f]];
}

#define DFX_SCAN_UNROLL 16
// Exponential mean norm of the ERB features (lib.rs:244-251) and exponential unit norm of the complex features
// (lib.rs:253-259) — true recurrences over time, so one thread owns one (row, channel) and walks T sequentially; the
// loads do not depend on the recurrence and are issued DFX_SCAN_UNROLL frames ahead.  Channels [0,E) are ERB bands,
// [E, E+Fn) complex bins.  Either half may be disabled by passing a null input pointer.
// erb_out_cs / spec_out_cs > 0: elements between the clips of erb_out / spec_out (outputs written straight into a window of a longer buffer:
// the streaming runtime's linear feature windows); 0: dense [C, T, E] / [C, T, Fn].
__global__ void dfx_k_norm_scan(const float *erb_in, float *erb_out, int E, const float2 *spec_in,
                                int64_t spec_frame_stride, float2 *spec_out, int Fn, int64_t C, int64_t T, float alpha,
                                float *erb_state, float *unit_state, int64_t erb_out_cs = 0, int64_t spec_out_cs = 0) {
#pragma clang fp contract(off)  // the Rust reference never fuses x*(1-a) + s*a into an FMA (lib.rs:244-259)
    const int nch = (erb_in ? E : 0) + (spec_in ? Fn : 0);
    const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= C * nch) return;
    const int64_t c = gid / nch;
    int ch = (int)(gid - c * nch);
    const float one_m_a = 1.f - alpha;
    if (erb_in && ch < E) {
        // state init: linspace(-60, -90, E) (transforms.rs:308-318; step form of ndarray::linspace)
        float s;
        if (erb_state) s = erb_state[c * E + ch];
        else s = -60.f + (E > 1 ? (-90.f - -60.f) / (float)(E - 1) : 0.f) * (float)ch;
        const float *in = erb_in + c * T * E + ch;
        float *out = erb_out + c * (erb_out_cs > 0 ? erb_out_cs : T * E) + ch;
        // batches of DFX_SCAN_UNROLL frames, double-buffered: batch k+1 is requested before batch k is scanned, so the recurrence
        // never waits for memory (a plain load-batch / scan-batch loop spends most of its time in the load latency)
        int64_t t = 0;
        const int64_t nbatch = T / DFX_SCAN_UNROLL;
        float v[DFX_SCAN_UNROLL] = {}, nv[DFX_SCAN_UNROLL] = {};
        if (nbatch > 0) {
#pragma unroll
            for (int u = 0; u < DFX_SCAN_UNROLL; ++u) v[u] = in[u * E];
        }
        for (int64_t bi = 0; bi < nbatch; ++bi, t += DFX_SCAN_UNROLL) {
            if (bi + 1 < nbatch) {
#pragma unroll
                for (int u = 0; u < DFX_SCAN_UNROLL; ++u) nv[u] = in[(t + DFX_SCAN_UNROLL + u) * E];
            }
#pragma unroll
            for (int u = 0; u < DFX_SCAN_UNROLL; ++u) {
                s = v[u] * one_m_a + s * alpha;
                out[(t + u) * E] = (v[u] - s) / 40.f;
            }
#pragma unroll
            for (int u = 0; u < DFX_SCAN_UNROLL; ++u) v[u] = nv[u];
        }
        for (; t < T; ++t) {
            const float v = in[t * E];
            s = v * one_m_a + s * alpha;
            out[t * E] = (v - s) / 40.f;
        }
        if (erb_state) erb_state[c * E + ch] = s;
        return;
    }
    if (erb_in) ch -= E;
    {
        float s;
        if (unit_state) s = unit_state[c * Fn + ch];
        else s = 0.001f + (Fn > 1 ? (0.0001f - 0.001f) / (float)(Fn - 1) : 0.f) * (float)ch;
        const float2 *in = spec_in + c * T * spec_frame_stride + ch;
        float2 *out = spec_out + c * (spec_out_cs > 0 ? spec_out_cs : T * Fn) + ch;
        int64_t t = 0;
        const int64_t nbatch = T / DFX_SCAN_UNROLL;
        float2 v[DFX_SCAN_UNROLL] = {}, nv[DFX_SCAN_UNROLL] = {};
        if (nbatch > 0) {
#pragma unroll
            for (int u = 0; u < DFX_SCAN_UNROLL; ++u) v[u] = in[u * spec_frame_stride];
        }
        for (int64_t bi = 0; bi < nbatch; ++bi, t += DFX_SCAN_UNROLL) {
            if (bi + 1 < nbatch) {
#pragma unroll
                for (int u = 0; u < DFX_SCAN_UNROLL; ++u) nv[u] = in[(t + DFX_SCAN_UNROLL + u) * spec_frame_stride];
            }
#pragma unroll
            for (int u = 0; u < DFX_SCAN_UNROLL; ++u) {
                s = hypotf(v[u].x, v[u].y) * one_m_a + s * alpha;
                const float d = sqrtf(s);
                out[(t + u) * Fn] = make_float2(v[u].x / d, v[u].y / d);
            }
#pragma unroll
            for (int u = 0; u < DFX_SCAN_UNROLL; ++u) v[u] = nv[u];
        }
        for (; t < T; ++t) {
            const float2 v = in[t * spec_frame_stride];
            s = hypotf(v.x, v.y) * one_m_a + s * alpha;
            const float d = sqrtf(s);
            out[t * Fn] = make_float2(v.x / d, v.y / d);
        }
        if (unit_state) unit_state[c * Fn + ch] = s;
    }
}

// The same scans with FOUR lanes per (row, channel): lane j of the quad owns the frames t = 4k + j.  Only the recurrence itself
// (s = x*(1-a) + s*a: two dependent operations per frame) is sequential; the magnitude (hypotf), the square root and the two divisions —
// ~50 of the ~55 instructions per frame — are not.  Every lane of the quad steps the recurrence through the quad's four frames (the
// magnitudes arrive by wave shuffle, so all four lanes hold the same s) and then finishes its own frame.  Same operations on the same
// values in the same order as dfx_k_norm_scan: same bits.  With one lane per channel a batch of 256 clips is 896 waves — less than one
// per SIMD, each issuing ~60 dependent-latency-bound instructions per frame: 0.26 ms for 0.46 GB; here 3584 waves of ~20 instructions per
// frame.
__global__ void __launch_bounds__(256) dfx_k_norm_scan4(const float *erb_in, float *erb_out, int E, const float2 *spec_in,
                                                        int64_t spec_frame_stride, float2 *spec_out, int Fn, int64_t C, int64_t T, float alpha,
                                                        float *erb_state, float *unit_state) {
#pragma clang fp contract(off)  // the Rust reference never fuses x*(1-a) + s*a into an FMA (lib.rs:244-259)
    constexpr int G = DFX_SCAN_UNROLL / 4;   // groups of 4 frames per batch
    const int nch = (erb_in ? E : 0) + (spec_in ? Fn : 0);
    const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int j = (int)(threadIdx.x & 3), qbase = (int)(threadIdx.x & 63) & ~3;
    const int64_t qid = gid >> 2;
    if (qid >= C * nch) return;      // (whole quads: the shuffles below stay inside a quad)
    const int64_t c = qid / nch;
    int ch = (int)(qid - c * nch);
    const float one_m_a = 1.f - alpha;
    if (erb_in && ch < E) {
        float s;
        if (erb_state) s = erb_state[c * E + ch];
        else s = -60.f + (E > 1 ? (-90.f - -60.f) / (float)(E - 1) : 0.f) * (float)ch;
        const float *in = erb_in + c * T * E + ch;
        float *out = erb_out + c * T * E + ch;
        int64_t t = 0;
        const int64_t nbatch = T / DFX_SCAN_UNROLL;
        float v[G] = {}, nv[G] = {};
        if (nbatch > 0) {
#pragma unroll
            for (int g = 0; g < G; ++g) v[g] = in[(4 * g + j) * E];
        }
        for (int64_t bi = 0; bi < nbatch; ++bi, t += DFX_SCAN_UNROLL) {
            if (bi + 1 < nbatch) {
#pragma unroll
                for (int g = 0; g < G; ++g) nv[g] = in[(t + DFX_SCAN_UNROLL + 4 * g + j) * E];
            }
#pragma unroll
            for (int g = 0; g < G; ++g) {
                float mine = 0.f;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const float xi = __shfl(v[g], qbase + i);
                    s = xi * one_m_a + s * alpha;
                    if (i == j) mine = s;
                }
                out[(t + 4 * g + j) * E] = (v[g] - mine) / 40.f;
            }
#pragma unroll
            for (int g = 0; g < G; ++g) v[g] = nv[g];
        }
        for (; t < T; t += 4) {   // tail groups (quad-uniform trip count)
            const bool have = t + j < T;
            const float x = have ? in[(t + j) * E] : 0.f;
            float mine = 0.f;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float xi = __shfl(x, qbase + i);
                if (t + i < T) {
                    s = xi * one_m_a + s * alpha;
                    if (i == j) mine = s;
                }
            }
            if (have) out[(t + j) * E] = (x - mine) / 40.f;
        }
        if (erb_state && j == 0) erb_state[c * E + ch] = s;
        return;
    }
    if (erb_in) ch -= E;
    {
        float s;
        if (unit_state) s = unit_state[c * Fn + ch];
        else s = 0.001f + (Fn > 1 ? (0.0001f - 0.001f) / (float)(Fn - 1) : 0.f) * (float)ch;
        const float2 *in = spec_in + c * T * spec_frame_stride + ch;
        float2 *out = spec_out + c * T * Fn + ch;
        int64_t t = 0;
        const int64_t nbatch = T / DFX_SCAN_UNROLL;
        float2 v[G] = {}, nv[G] = {};
        if (nbatch > 0) {
#pragma unroll
            for (int g = 0; g < G; ++g) v[g] = in[(4 * g + j) * spec_frame_stride];
        }
        for (int64_t bi = 0; bi < nbatch; ++bi, t += DFX_SCAN_UNROLL) {
            if (bi + 1 < nbatch) {
#pragma unroll
                for (int g = 0; g < G; ++g) nv[g] = in[(t + DFX_SCAN_UNROLL + 4 * g + j) * spec_frame_stride];
            }
#pragma unroll
            for (int g = 0; g < G; ++g) {
                const float h = hypotf(v[g].x, v[g].y);
                float mine = 0.f;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const float hi = __shfl(h, qbase + i);
                    s = hi * one_m_a + s * alpha;
                    if (i == j) mine = s;
                }
                const float d = sqrtf(mine);
                out[(t + 4 * g + j) * Fn] = make_float2(v[g].x / d, v[g].y / d);
            }
#pragma unroll
            for (int g = 0; g < G; ++g) v[g] = nv[g];
        }
        for (; t < T; t += 4) {
            const bool have = t + j < T;
            const float2 x = have ? in[(t + j) * spec_frame_stride] : make_float2(0.f, 0.f);
            const float h = hypotf(x.x, x.y);
            float mine = 1.f;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float hi = __shfl(h, qbase + i);
                if (t + i < T) {
                    s = hi * one_m_a + s * alpha;
                    if (i == j) mine = s;
                }
            }
            if (have) {
                const float d = sqrtf(mine);
                out[(t + j) * Fn] = make_float2(x.x / d, x.y / d);
            }
        }
        if (unit_state && j == 0) unit_state[c * Fn + ch] = s;
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// Fused deep filtering + ERB gains (+ post filter, + attenuation limit): the memory-bound kernel with the explicit
// >= 70 % HBM roofline target.  Algorithmic traffic per frame (O taps, nb_df bins, F bins, E bands):
//   read X  F*8, read coefs nb_df*O*8, read gains E*4, write Y F*8   (11 664 B for O=5, nb_df=96, F=481, E=32).
//
// Pure streaming design, no LDS, no barrier: a workgroup owns DFX_DFA_ROWS consecutive frames of one clip and walks
// their [rows*F] complex values as ONE flat 16-byte stream (F is odd, so frame rows are only 8-byte aligned: the flat view
// keeps every global access of the dominant streams a coalesced float4; at most one element is peeled at each end).
//   lane i: float4 = bins (f, f+1) of a frame (or the last bin of one frame and the first of the next)
//     f >= nb_df : y = x * gain[band(f)]                    (Mask, modules.py:266-269 == lib.rs:314-326)
//     f <  nb_df : y = sum_n C[n][f] * X[t+n-(O-1-la)][f]   (MF.DF, multiframe.py:126-136,169-180) with the taps and the
//                  coefficients loaded directly (8-byte loads; with the tap-major layouts consecutive lanes read consecutive
//                  coefficients, and the tap rows of X are L2 hits: the clip -> XCD mapping keeps a clip on one L2)
// The coefficient layout is described by four strides (complex elements): BOTF, BTFO and the engine's own BTOF.
// ---------------------------------------------------------------------------------------------------------------------
#define DFX_DFA_ROWS 16  // rows per workgroup
#define DFX_DFA_THREADS 256

struct DfxDfaArgs {
    const float2 *spec;   // [B,T,F]
    const float2 *coefs;  // element (b,t,n,f) at b*cs_b + t*cs_t + n*cs_n + f*cs_f
    const float *gains;   // [B,T,nb] or null
    const unsigned char *bin2band;  // [F]
    float2 *out;          // [B,T,F]
    int64_t B, T;
    int64_t cs_b, cs_t, cs_n, cs_f;
    int F, nbdf, order, lookahead, nb;
    float pf_beta, atten_lim;
    int chunks;           // row chunks per clip
    int t_begin, t_end;   // frames [t_begin, t_end) of every clip are produced by this launch
    int64_t gT;           // frames per clip of the gains array (== T except for streaming windows, whose spec carries extra lookahead frames)
    int64_t out_T, out_toff;  // frame t of clip b is stored at out[(b*out_T + t - out_toff)*F ...]  (== T, 0 unless compacted)
    // post filter of the real-time runtime (DfTract::process -> lib.rs:446-471): pf_ch > 0 selects the Rust arithmetic and its
    // chunks_exact(4) walk over the frame's [pf_ch * F] bins (the pf_ch rows of a stream are consecutive clips): the last
    // (pf_ch * F) % 4 bins of the flattened frame are NOT filtered (mono, F = 481: bin 480).  0: deepfilternet3.py:448-454.
    int pf_ch = 0;
};

// lib.rs:446-471 post_filter, one bin: g = min(|e| / (|n| + eps), 1).max(eps); g_sin = g * sin(g * pi / 2);
// pf = (beta_p1 * g / (1 + beta * (g / g_sin)^2)) / g   — the reference's own order of operations, no clamp on the sine
static __device__ __forceinline__ float2 dfx_dfa_post_filter_rs(float2 y, float2 x, float beta) {
#pragma clang fp contract(off)
    const float eps = 1e-12f, pi = 3.14159265358979323846f, beta_p1 = beta + 1.f;
    float g = hypotf(y.x, y.y) / (hypotf(x.x, x.y) + eps);
    g = fmaxf(fminf(g, 1.f), eps);
    const float g_sin = g * sinf(g * pi / 2.0f);
    const float q = g / g_sin;
    const float pf = (beta_p1 * g / (1.f + beta * (q * q))) / g;
    return make_float2(y.x * pf, y.y * pf);
}

static __device__ __forceinline__ float2 dfx_dfa_finish(float2 y, float2 x, float pf_beta, float lim) {
    if (pf_beta > 0.f) {
        // deepfilternet3.py:448-454 (== lib.rs:446-471)
        const float eps = 1e-12f, pi = 3.14159265358979323846f;
        float g = sqrtf(y.x * y.x + y.y * y.y) / (sqrtf(x.x * x.x + x.y * x.y) + eps);
        g = fminf(fmaxf(g, eps), 1.f);
        const float g_sin = g * fmaxf(sinf(pi * g * 0.5f), eps);
        const float q = g / g_sin;
        const float pf = (1.f + pf_beta) / (1.f + pf_beta * q * q);
        y.x *= pf;
        y.y *= pf;
    }
    if (lim > 0.f) {
        // enhance.py:238-240
        y.x = x.x * lim + y.x * (1.f - lim);
        y.y = x.y * lim + y.y * (1.f - lim);
    }
    return y;
}

// one output bin of frame t (row index inside the clip), bin f, x = spec[b,t,f].  xs: the low bins [0,nbdf) of frames
// t0-toff .. of this chunk staged in LDS (zeros outside the clip); gs / b2b: the chunk's gains and the bin->band map in LDS.
static __device__ __forceinline__ float2 dfx_dfa_bin(const DfxDfaArgs &A, const float2 *xs, const float *gs,
                                                     const unsigned char *b2b, const float2 *coef_b, int t0, int t, int f,
                                                     float2 x, int ch_off = 0) {
    float2 y;
    if (f < A.nbdf) {
        float re = 0.f, im = 0.f;
        const float2 *cp = coef_b + (int64_t)t * A.cs_t + (int64_t)f * A.cs_f;
        const float2 *xp = xs + (t - t0) * A.nbdf + f;  // tap n reads frame t + n - toff == staged row (t - t0) + n
        for (int n = 0; n < A.order; ++n) {
            const float2 c = cp[(int64_t)n * A.cs_n];
            const float2 xx = xp[n * A.nbdf];
            re += xx.x * c.x - xx.y * c.y;
            im += xx.x * c.y + xx.y * c.x;
        }
        y = make_float2(re, im);
    } else if (gs) {
        const float g = gs[(t - t0) * A.nb + b2b[f]];
        y = make_float2(x.x * g, x.y * g);
    } else {
        y = x;
    }
    if (A.pf_ch > 0) {   // the runtime's post filter: Rust arithmetic, the tail of the flattened [pf_ch * F] frame left alone
        if (A.pf_beta > 0.f && ch_off + f < ((A.pf_ch * A.F) & ~3)) y = dfx_dfa_post_filter_rs(y, x, A.pf_beta);
        return dfx_dfa_finish(y, x, 0.f, A.atten_lim);
    }
    return dfx_dfa_finish(y, x, A.pf_beta, A.atten_lim);
}

template <int ROWS>
__global__ void __launch_bounds__(DFX_DFA_THREADS) dfx_k_df_apply(DfxDfaArgs A) {
    DFX_DYN_SMEM(unsigned char, smem);
    // blocks that share blockIdx.x % 8 (one XCD, one L2) work on the same clips (placement is a speed hint only)
    const int64_t id = blockIdx.x;
    const int xcd = (int)(id & 7);
    const int64_t j = id >> 3;
    const int chunk = (int)(j % A.chunks);
    const int64_t b = (j / A.chunks) * 8 + xcd;
    if (b >= A.B) return;
    const int F = A.F, nd = A.nbdf;
    const int ch_off = A.pf_ch > 0 ? (int)(b % A.pf_ch) * F : 0;   // position of this row's bin 0 in its stream's flattened frame
    const int t0 = A.t_begin + chunk * ROWS;
    const int nt = (A.t_end - t0) < ROWS ? (A.t_end - t0) : ROWS;
    const int halo = ROWS + A.order - 1;
    float2 *xs = reinterpret_cast<float2 *>(smem);                                   // [halo][nd]
    size_t off = ((size_t)halo * nd * 8 + 15) & ~(size_t)15;
    float *gs = reinterpret_cast<float *>(smem + off);                               // [ROWS][nb]
    off += ((size_t)ROWS * (A.nb > 0 ? A.nb : 1) * 4 + 15) & ~(size_t)15;
    unsigned char *b2b = smem + off;                                                 // [F]
    const float2 *spec_b = A.spec + b * A.T * F;
    float2 *out_b = A.out + (b * A.out_T - A.out_toff) * F;
    const float2 *coef_b = A.coefs + b * A.cs_b;
    const int tid = threadIdx.x;
    // ---- stage the deep filter's input window: every low bin of the chunk (+ order-1 halo frames) is read from HBM once
    const int toff = A.order - 1 - A.lookahead;
    for (int i = tid; i < halo * nd; i += DFX_DFA_THREADS) {
        const int h = i / nd, f = i - h * nd;
        const int tt = t0 - toff + h;
        float2 v = make_float2(0.f, 0.f);
        if (tt >= 0 && tt < A.T) v = spec_b[(int64_t)tt * F + f];
        xs[i] = v;
    }
    if (A.gains) {
        const float *gp = A.gains + (b * A.gT + t0) * A.nb;
        for (int i = tid; i < nt * A.nb; i += DFX_DFA_THREADS) gs[i] = gp[i];
        for (int i = tid; i < F; i += DFX_DFA_THREADS) b2b[i] = A.bin2band[i];
    } else {
        gs = nullptr;
    }
    __syncthreads();
    // ---- flat stream over the chunk's [nt*F] bins: a head and a tail of < 32 bins are peeled so that the float4 body
    // starts and ends on 256-byte boundaries (every wave-wide access then covers whole cache lines)
    const int e0 = t0 * F, e1 = e0 + nt * F;
    const int mis = (int)((reinterpret_cast<uintptr_t>(spec_b + e0) >> 3) & 31);   // bins past a 256-byte boundary
    int a0 = e0 + ((32 - mis) & 31);
    if (a0 > e1) a0 = e1;
    const int a1 = a0 + ((e1 - a0) & ~31);
    for (int e = e0 + tid; e < a0; e += DFX_DFA_THREADS) {
        const int t = e / F, f = e - t * F;
        const float2 x = (f < nd) ? xs[(t - t0 + toff) * nd + f] : spec_b[e];
        out_b[e] = dfx_dfa_bin(A, xs, gs, b2b, coef_b, t0, t, f, x, ch_off);
    }
    for (int e = a1 + tid; e < e1; e += DFX_DFA_THREADS) {
        const int t = e / F, f = e - t * F;
        const float2 x = (f < nd) ? xs[(t - t0 + toff) * nd + f] : spec_b[e];
        out_b[e] = dfx_dfa_bin(A, xs, gs, b2b, coef_b, t0, t, f, x, ch_off);
    }
    const float4 *x4 = reinterpret_cast<const float4 *>(spec_b + a0);
    float4 *y4 = reinterpret_cast<float4 *>(out_b + a0);
    const int n4 = (a1 - a0) >> 1;
    // two float4s per iteration: both loads are issued before either is used
    for (int i = tid; i < n4; i += 2 * DFX_DFA_THREADS) {
        const int i2 = i + DFX_DFA_THREADS;
        const bool has2 = i2 < n4;
        int tt[2], ff[2], tt2[2], ff2[2];
        bool lds[2];
        float4 xv[2];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int e = a0 + 2 * (u ? i2 : i);
            tt[u] = e / F;
            ff[u] = e - tt[u] * F;
            tt2[u] = tt[u];
            ff2[u] = ff[u] + 1;
            if (ff2[u] == F) {
                ff2[u] = 0;
                ++tt2[u];
            }
            lds[u] = ff2[u] < nd && ff[u] < nd && ff2[u] != 0;  // both bins belong to the deep filter: inputs are in LDS
        }
        if (!lds[0]) xv[0] = x4[i];
        if (has2 && !lds[1]) xv[1] = x4[i2];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            if (u == 1 && !has2) break;
            if (lds[u]) {
                const float2 xa = xs[(tt[u] - t0 + toff) * nd + ff[u]], xb = xs[(tt[u] - t0 + toff) * nd + ff2[u]];
                xv[u] = make_float4(xa.x, xa.y, xb.x, xb.y);
            }
            const float2 ya = dfx_dfa_bin(A, xs, gs, b2b, coef_b, t0, tt[u], ff[u], make_float2(xv[u].x, xv[u].y), ch_off);
            const float2 yb = dfx_dfa_bin(A, xs, gs, b2b, coef_b, t0, tt2[u], ff2[u], make_float2(xv[u].z, xv[u].w), ch_off);
            y4[u ? i2 : i] = make_float4(ya.x, ya.y, yb.x, yb.y);
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// Row-streaming form of the same operator for 16-byte aligned rows (row stride Fs complex elements, Fs even: the engine pads
// F = 481 to 482 in its own spec / spec_e buffers, so a row is a run of float4 = two bins and nothing straddles rows).
//   * a WAVE owns `rpw` consecutive frames of one clip (default 1: measured fastest, 6.2 TB/s vs 5.9 at 4 and 5.4 at 16) and walks
//     them in time order; a workgroup = 4 waves = 4 consecutive chunks of that clip.  No LDS, no barrier: waves never wait for
//     each other.  All streams are non-temporal (every byte is touched once; a plain copy gains 10 % from the hint on this part).
//   * pass 0 (lanes 0..63 = bins 0..127): lanes below nb_df/2 own two deep-filter bins each and keep the O frames the filter
//     reads in a REGISTER ring (one new 16-byte load per frame, not O); their O coefficients are O float4 loads (tap-major or
//     frame-major layouts: consecutive lanes read consecutive coefficients).  The remaining lanes / passes multiply by the band
//     gain; a frame's E gains are one coalesced load (lane e holds band e) + one wave shuffle per bin (the bands of a lane's
//     bins never change, so their indices live in registers).
//   * every global access is a whole-row, 16-byte-per-lane access; the pad bin of a row is written as zero.
// HBM traffic = the algorithmic bytes + the O-1 ring-fill frames per chunk (low bins only: (O-1)/rpw * nb_df*8 B per frame).
// ---------------------------------------------------------------------------------------------------------------------
struct DfxDfrArgs {
    const float *spec;    // [B, T, Fs][2]
    const float *coefs;   // complex element (b,t,n,f) at b*cs_b + t*cs_t + n*cs_n + f  (cs_* even: float4 aligned)
    const float *gains;   // [B, gT, nb] or null
    const unsigned char *bin2band;  // [F]
    float *out;           // [B, out_T, Fso][2]
    int64_t B, T;
    int64_t cs_b, cs_t, cs_n;
    int64_t gT, out_T, out_toff;
    int Fs, Fso;          // row strides in complex elements (even)
    int F, nbdf, lookahead, nb;
    float pf_beta, atten_lim;
    int t_begin, t_end;   // frames [t_begin, t_end) of every clip
    int rpw;              // frames per wave
    int zcols;            // float4 columns stored per output row: ceil(F/2), or more (zeros) to complete the row's last 64-byte sector
    int chunks;           // chunks of rpw frames per clip
    int64_t items;        // work items = ceil(B / 8) * 8 * ceil(chunks / 4)
    int pf_ch = 0;        // > 0: the real-time runtime's post filter (see DfxDfaArgs::pf_ch): pf_ch consecutive clips are the channels of one stream
};
// A pass of enhance() whose kernels raised a fault (err[1]: fp16-split range, err[2]: a flag wait timed out and its workgroup ran on without its data)
// must not hand back plausible samples: the finishing kernel looks at the words when it starts — every fault that can reach its inputs has been
// raised by then, it runs behind everything else of the pass — and stores NaN (16-bit PCM: zeros) instead.  The host still gets the error from the
// words.  The words live in host memory — a read over PCIe each: every thread reading them cost the kernel 1.7 ms, one lane per workgroup still
// 0.12 ms — so a one-thread launch in front of the finishing kernel copies their verdict into a word of device memory (dfx_k_fault_mirror), and
// that is what the finishing kernel reads.
__global__ void dfx_k_fault_mirror(const unsigned int *err, unsigned int *poison) {
    *poison = (__hip_atomic_load(err + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) | __hip_atomic_load(err + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM)) != 0u ? 1u : 0u;
}

// NPC > 0: the number of 64-lane passes over a row (ceil(ceil(F/2) / 64)) as a compile-time constant (all row loads of a frame are
// then issued together and the band indices live in registers); NPC == 0: any F, one pass at a time.  PF: post filter / attenuation
// limit compiled in (the common case, neither, then carries no sinf / sqrtf code and fewer live registers).
// Addressing: everything but the lane index is wave-uniform (clip, frame, tap), so every access is <scalar base> + lane * 16.
template <int O, int NPC, bool PF, int NT = 7>   // NT bit 0: non-temporal spec loads, bit 1: non-temporal stores, bit 2: non-temporal coefficient loads
__global__ void __launch_bounds__(256) dfx_k_df_apply_rows(DfxDfrArgs A) {
    auto ld = [](const f32x4 *p) -> f32x4 { return (NT & 1) ? DFX_NT_LOAD(p) : *p; };
    auto ldc = [](const f32x4 *p) -> f32x4 { return (NT & 4) ? DFX_NT_LOAD(p) : *p; };
    auto st = [](f32x4 v, f32x4 *p) { if (NT & 2) DFX_NT_STORE(v, p); else *p = v; };
    const unsigned lane = threadIdx.x & 63;
    const int wave = dfx_wave_uniform((int)(threadIdx.x >> 6));
    const int wgc = (A.chunks + 3) >> 2;    // work items (4 chunks = one workgroup pass) per clip
    // the grid is a multiple of 8 and walks the work items grid-stride; by default it has one workgroup per item (a persistent grid of
    // a few workgroups per CU was measured slower, 0.54 vs 0.47 ms: short-lived workgroups keep more independent rows in flight)
    for (int64_t id = blockIdx.x; id < A.items; id += gridDim.x) {
    const int xcd = (int)(id & 7);          // items that share id % 8 run on one XCD / L2 (gridDim % 8 == 0) and work on the same clips
    const int64_t j = id >> 3;
    const int64_t b = (j / wgc) * 8 + xcd;
    const int chunk = (int)(j % wgc) * 4 + wave;
    if (b >= A.B || chunk >= A.chunks) continue;
    const int t0 = A.t_begin + chunk * A.rpw;
    const int t1 = (t0 + A.rpw) < A.t_end ? (t0 + A.rpw) : A.t_end;
    const int F = A.F, la = A.lookahead, toff = O - 1 - la;
    const unsigned nd4 = (unsigned)A.nbdf >> 1;   // float4 columns that belong to the deep filter (<= 64)
    const unsigned ncol = (unsigned)(F + 1) >> 1; // float4 columns of a row
    const unsigned zcol = (unsigned)A.zcols;      // columns written per row (>= ncol: the extra ones are zero padding)
    const int np = NPC > 0 ? NPC : (int)(((zcol > ncol ? zcol : ncol) + 63) >> 6);
    const bool is_df = lane < nd4;
    const int64_t rs = A.Fs >> 1, ro = A.Fso >> 1, cst = A.cs_t >> 1, csn = A.cs_n >> 1;
    const f32x4 *spec_b = reinterpret_cast<const f32x4 *>(A.spec) + b * A.T * rs;
    const f32x4 *coef_b = reinterpret_cast<const f32x4 *>(A.coefs) + ((b * A.cs_b) >> 1);
    f32x4 *out_b = reinterpret_cast<f32x4 *>(A.out) + (b * A.out_T - A.out_toff) * ro;
    const float *gain_b = A.gains ? A.gains + b * A.gT * A.nb : nullptr;
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
    // band indices of this lane's bins (pass p: bins 2*(lane + 64p), +1); a bin past F-1 is the row's pad (written as 0)
    constexpr int NB = NPC > 0 ? NPC : 1;
    int band0[NB], band1[NB];
    if (NPC > 0 && A.gains) {
#pragma unroll
        for (int p = 0; p < NB; ++p) {
            const int f = 2 * ((int)lane + 64 * p);
            band0[p] = f < F ? A.bin2band[f] : 0;
            band1[p] = f + 1 < F ? A.bin2band[f + 1] : 0;
        }
    }
    // (pf_ch > 0: libDF's post filter — Rust arithmetic, and the last (pf_ch * F) % 4 bins of the stream's flattened [pf_ch * F] frame are
    // left alone, lib.rs:446-471 — then the attenuation limit; see dfx_dfa_bin)
    const int pf_lim4 = (A.pf_ch * F) & ~3, ch_off = A.pf_ch > 0 ? (int)(b % A.pf_ch) * F : 0;
    auto finish2 = [&](f32x4 y, f32x4 x, bool second_is_pad, int f0) -> f32x4 {
        if (PF) {
            float2 ya = make_float2(y[0], y[1]), yb = make_float2(y[2], y[3]);
            const float2 xa = make_float2(x[0], x[1]), xb = make_float2(x[2], x[3]);
            if (A.pf_ch > 0) {
                if (A.pf_beta > 0.f && ch_off + f0 < pf_lim4) ya = dfx_dfa_post_filter_rs(ya, xa, A.pf_beta);
                if (A.pf_beta > 0.f && ch_off + f0 + 1 < pf_lim4) yb = dfx_dfa_post_filter_rs(yb, xb, A.pf_beta);
                ya = dfx_dfa_finish(ya, xa, 0.f, A.atten_lim), yb = dfx_dfa_finish(yb, xb, 0.f, A.atten_lim);
            } else {
                ya = dfx_dfa_finish(ya, xa, A.pf_beta, A.atten_lim), yb = dfx_dfa_finish(yb, xb, A.pf_beta, A.atten_lim);
            }
            y[0] = ya.x, y[1] = ya.y, y[2] = yb.x, y[3] = yb.y;
        }
        if (second_is_pad) y[2] = 0.f, y[3] = 0.f;
        return y;
    };
    // ring[n] = frame t - toff + n of this lane's two deep-filter bins
    f32x4 ring[O];
#pragma unroll
    for (int n = 0; n < O; ++n) ring[n] = zero4;
    if (is_df) {
#pragma unroll
        for (int n = 0; n + 1 < O; ++n) {
            const int tt = t0 - toff + n;
            if (tt >= 0 && tt < A.T) ring[n + 1] = (spec_b + (int64_t)tt * rs)[lane];   // re-read by the neighbouring chunk: default policy
        }
    }
    for (int t = t0; t < t1; ++t) {
        const f32x4 *xrow = spec_b + (int64_t)t * rs;
        f32x4 *yrow = out_b + (int64_t)t * ro;
        // ---- issue the frame's loads: pass 0 (the ring's newest frame for the deep-filter lanes, the frame itself otherwise),
        // the coefficients, the gains, then the remaining passes
        f32x4 x0 = zero4;
        if (is_df) {
            if (t + la < A.T) x0 = ld(xrow + (int64_t)la * rs + lane);
        } else if (lane < ncol) {
            x0 = ld(xrow + lane);
        }
        f32x4 cf[O];
        if (is_df) {
            const f32x4 *cp = coef_b + (int64_t)t * cst;
#pragma unroll
            for (int n = 0; n < O; ++n) cf[n] = ldc(cp + (int64_t)n * csn + lane);
        }
        float gv = 1.f;
        if (gain_b && lane < (unsigned)A.nb) gv = (gain_b + (int64_t)t * A.nb)[lane];
        f32x4 xp[NB];
        if (NPC > 0) {
#pragma unroll
            for (int p = 1; p < NB; ++p) xp[p] = (lane + 64u * p) < ncol ? ld(xrow + 64 * p + lane) : zero4;
        }
        // ---- pass 0
        {
            f32x4 x = x0, y;
            if (is_df) {
#pragma unroll
                for (int n = 0; n + 1 < O; ++n) ring[n] = ring[n + 1];
                ring[O - 1] = x0;
                if (PF) {  // the frame itself (ring slot toff), selected without a dynamically indexed register array
#pragma unroll
                    for (int n = 0; n < O; ++n)
                        if (n == toff) x = ring[n];
                }
                y = zero4;
#pragma unroll
                for (int n = 0; n < O; ++n) {
                    const f32x4 c = cf[n], xx = ring[n];
                    y[0] += xx[0] * c[0] - xx[1] * c[1];
                    y[1] += xx[0] * c[1] + xx[1] * c[0];
                    y[2] += xx[2] * c[2] - xx[3] * c[3];
                    y[3] += xx[2] * c[3] + xx[3] * c[2];
                }
            }
            float g0 = 1.f, g1 = 1.f;
            if (A.gains) {  // every lane takes part in the shuffles
                int i0, i1;
                if (NPC > 0) {
                    i0 = band0[0], i1 = band1[0];
                } else {
                    const int f = 2 * (int)lane;
                    i0 = f < F ? A.bin2band[f] : 0;
                    i1 = f + 1 < F ? A.bin2band[f + 1] : 0;
                }
                g0 = __shfl(gv, i0);
                g1 = __shfl(gv, i1);
            }
            if (!is_df) {
                y[0] = x[0] * g0, y[1] = x[1] * g0;
                y[2] = x[2] * g1, y[3] = x[3] * g1;
            }
            if (lane < ncol) st(finish2(y, x, 2 * (int)lane + 1 >= F, 2 * (int)lane), yrow + lane);
            else if (lane < zcol) st(zero4, yrow + lane);
        }
        // ---- the other passes: band gains only
        auto gain_pass = [&](int p, f32x4 x, int i0, int i1) {
            const unsigned col = lane + 64u * p;
            float g0 = 1.f, g1 = 1.f;
            if (A.gains) {
                g0 = __shfl(gv, i0);
                g1 = __shfl(gv, i1);
            }
            if (col < ncol) {
                f32x4 y;
                y[0] = x[0] * g0, y[1] = x[1] * g0;
                y[2] = x[2] * g1, y[3] = x[3] * g1;
                st(finish2(y, x, 2 * (int)col + 1 >= F, 2 * (int)col), yrow + 64 * p + lane);
            } else if (col < zcol) {
                st(zero4, yrow + 64 * p + lane);   // pad columns that complete the row's last 64-byte sector
            }
        };
        if (NPC > 0) {
#pragma unroll
            for (int p = 1; p < NB; ++p) gain_pass(p, xp[p], A.gains ? band0[p] : 0, A.gains ? band1[p] : 0);
        } else {
            for (int p = 1; p < np; ++p) {
                const unsigned col = lane + 64u * p;
                const int f = 2 * (int)col;
                const f32x4 x = col < ncol ? ld(xrow + 64 * p + lane) : zero4;
                gain_pass(p, x, (A.gains && f < F) ? A.bin2band[f] : 0, (A.gains && f + 1 < F) ? A.bin2band[f + 1] : 0);
            }
        }
    }
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// dfx_k_synthesis_rows<O, PF>: the finishing pass of enhance() for the 48 kHz / 20 ms configuration (N = 960, hop = 480, the in-place
// 480-point transform) in ONE kernel — Mask + MF.DF + combine [+ post filter + attenuation limit] (multiframe.py:126-180,
// deepfilternet3.py:426-454, enhance.py:238-240: the arithmetic of dfx_k_df_apply_rows) applied while a frame is on its way INTO the
// inverse transform, then ISTFT + window + overlap-add (lib.rs:396-427: dfx_k_synthesis<true>).  The enhanced spectrum never exists in
// HBM: per frame the pass reads X (3848 B), the O taps' coefficients (O * nb_df * 8 B) and the band gains (128 B), and writes 1920 B of
// audio — 9736 algorithmic bytes instead of 11664 + 5768 for the two kernels (the O - 1 neighbouring frames a frame's taps read are
// L2 hits: the neighbouring waves of the workgroup load the same rows at the same time).
// Work: a workgroup walks a SEGMENT of consecutive 8-frame chunks of one row and carries the second half of a chunk's last frame to the
// next chunk in LDS (double buffered); all 8 frames of a chunk are output frames (dfx_k_synthesis recomputes one halo frame per 7).
// A segment that does not start at frame 0 begins with one prologue item in which a single wave transforms the frame before it.
// O = 0: the input is the enhanced spectrum itself (no deep filter: the carry alone).  Sums in the reference's order (older frame first).
// ---------------------------------------------------------------------------------------------------------------------
struct DfxSynRowsArgs {
    const float2 *spec;       // [B, Tf, spec_stride]: O > 0 the noisy spectrum X, O == 0 the enhanced spectrum
    const float2 *coefs;      // O > 0: complex element (b, n, t, f) at b * cs_b + n * cs_n + t * cs_t + f
    const float *gains;       // [B, Tf, nb] or null (O > 0)
    const unsigned char *bin2band;   // [F]
    float *out;               // [B, out_stride] (I16 instances: int16_t samples behind the same pointer)
    const float *window;      // [960]
    const float2 *tw;         // [960]
    int64_t B, Tf, spec_stride, out_stride, out_skip, out_len;
    int64_t cs_b, cs_n, cs_t;
    int nbdf, lookahead, nb;
    float pf_beta, atten_lim;
    int segs, seg_chunks;     // segments per row, 8-frame chunks per segment
    const unsigned char *mfft = nullptr;   // MF instances: the inverse tables of dfx_fft480_mfma
    const unsigned int *poison = nullptr;  // device word, non-zero: a kernel of this pass raised a fault — store NaN (16-bit PCM: zeros); dfx_k_fault_mirror
};
// eight frames per chunk = eight waves per workgroup, three workgroups per CU (six waves per SIMD).  Round 6 measured the other shapes the LDS admits
// (same bits): 4 waves x 5 workgroups 0.66 ms, 6 x 4 0.61, 4 x 4 0.54 against 0.52 — the kernel wants FEW, LONG streams (a chunk reads 31 KB of
// spectrum rows and 6 KB of each tap's coefficients in one piece); profiles/r06_stft_variants.log
#define DFX_SYNR_TEAMS 8
#define DFX_SYNR_WPS 6
#define DFX_SYNR_WGS 3
#define DFX_SYNR_THREADS (DFX_DSP_TEAM * DFX_SYNR_TEAMS)
#define DFX_SYNR_TAB 6560  /* per-pass twiddle tables (DFX_TW480_BYTES) + the pre-pass factors [241] */
#define DFX_SYNR_SMEM ((size_t)DFX_SYNR_TAB + 960 * 4 + (size_t)DFX_SYNR_TEAMS * DFX_FFT480_BUF * 8 + (size_t)2 * 480 * 4 + 496)
#define DFX_SYNR_SMEM_MF ((size_t)960 * 12 + (size_t)DFX_SYNR_TEAMS * DFX_FFT480_BUF * 8 + (size_t)2 * 480 * 4 + 496 + (size_t)DFX_MFFT_FRAG3 * 64 * 16)

template <int O, bool PF, bool I16 = false, bool MF = false>
__global__ void __launch_bounds__(DFX_SYNR_THREADS, MF ? 4 : (O > 0 ? 4 : DFX_SYNR_WPS)) dfx_k_synthesis_rows(DfxSynRowsArgs A) {   // (O > 0: the next frame's fifteen loads stay in flight across the item: ~60 registers more)
    constexpr int M = 480, N = 960, HOP = 480, NTM = DFX_SYNR_TEAMS, BUF = DFX_FFT480_BUF;
    constexpr size_t TAB = MF ? (size_t)N * 8 : (size_t)DFX_SYNR_TAB;
    DFX_DYN_SMEM(unsigned char, smem);
    float2 *tw = reinterpret_cast<float2 *>(smem);
    float *win = reinterpret_cast<float *>(smem + TAB);
    dfx_h8 *a3s = reinterpret_cast<dfx_h8 *>(smem + TAB + (size_t)N * 4);    // [8][64] (MF)
    float2 *bufs = reinterpret_cast<float2 *>(smem + TAB + (size_t)N * 4 + (MF ? (size_t)DFX_MFFT_FRAG3 * 64 * 16 : 0));
    DfxMfftRegs mfr;
    if constexpr (MF) {
        for (int i = threadIdx.x; i < DFX_MFFT_FRAG3 * 64; i += DFX_SYNR_THREADS) a3s[i] = reinterpret_cast<const dfx_h8 *>(A.mfft)[DFX_MFFT_FRAG1 * 64 + i];
        dfx_mfft_load(mfr, A.mfft, (int)(threadIdx.x % DFX_DSP_TEAM));
    }
    float *carry = reinterpret_cast<float *>(bufs + (size_t)NTM * BUF);   // [2][HOP]
    unsigned char *b2b = reinterpret_cast<unsigned char *>(carry + 2 * HOP);   // [M + 1] (+ pad)
    const int team = dfx_wave_uniform(threadIdx.x / DFX_DSP_TEAM), lane = threadIdx.x % DFX_DSP_TEAM;
    float2 *bufA = bufs + (size_t)team * BUF;
    const bool poisoned = A.poison && *A.poison != 0u;   // (a faulted pass: the window table carries NaN, so every sample of the pass does)
    // T480: the table area holds the per-pass tables of dfx_fft480_t and, behind them, the pre-pass factors tp[k] = exp(-2 pi i k / N), k <= M / 2
    constexpr bool T480 = !MF;
    f32x4 *t1 = reinterpret_cast<f32x4 *>(smem), *t2 = t1 + DFX_TW480_T1;
    float2 *tp = T480 ? reinterpret_cast<float2 *>(smem + DFX_TW480_BYTES) : tw;   // [M / 2 + 1]
    for (int i0 = threadIdx.x; i0 < N; i0 += 4 * DFX_SYNR_THREADS) {   // all loads of a pass before its first LDS store
        float2 tv[4];
        float wv[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int i = i0 + u * DFX_SYNR_THREADS;
            tv[u] = i < N && (!T480 || i <= M / 2) ? A.tw[i] : make_float2(0.f, 0.f);
            wv[u] = i < N ? (poisoned ? __builtin_nanf("") : A.window[i]) : 0.f;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int i = i0 + u * DFX_SYNR_THREADS;
            if (i < N) {
                if (!T480 || i <= M / 2) tp[i] = tv[u];
                win[i] = wv[u];
            }
        }
    }
    if constexpr (T480) dfx_tw480_fill(t1, t2, A.tw, (int)threadIdx.x, DFX_SYNR_THREADS);
    // the bands of this lane's eight bins 2 l + 128 u (+ 1), each as the byte address 4 * band of the lane that holds the band's gain: they depend
    // on nothing but the lane (round 6: were two LDS byte reads + mask + shift in front of every gather)
    unsigned bandq[2] = {0u, 0u};
    if (O > 0 && A.gains) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int k = 2 * lane + 128 * u;
            const unsigned b0 = A.bin2band[k <= M ? k : 0], b1 = A.bin2band[k + 1 <= M ? k + 1 : 0];
            bandq[u >> 1] |= (((b0 & 63u) << 2) | ((b1 & 63u) << 10)) << (16 * (u & 1));
        }
    }
    (void)b2b;
    __syncthreads();
    const int toff = O - 1 - A.lookahead;
    const int64_t chunks = (A.Tf + NTM - 1) / NTM;
    int par = 0;   // carry buffer the current item reads
    // ---- The items of this workgroup — for every (row, segment) it is dealt: [one prologue item,] the segment's chunks — as ONE software-pipelined loop
    // (round 6): the loads of item i + 1 are requested as soon as item i's values have left their registers for the LDS, and are on their way while
    // item i goes through pre-pass, transform, overlap-add and stores.  Round-6 ablations (profiles/r06_finish_ablation.log): without the transform
    // the kernel took 7 % less, without its stores 18 %, with a fifth of its loads 25 % — bound by the bytes it has in flight (fifteen 16-byte loads
    // per lane, but only for the quarter of a frame's time that its wave waited for them), not by its instructions.  The frame in flight lives in
    // ~60 registers across the whole item: four waves per SIMD (two workgroups per CU) instead of six.
    struct Item {
        int64_t seg, b, c0, c1, ch;
    };
    const int64_t nseg = A.B * A.segs;
    auto start_seg = [&](Item &I) {
        I.b = I.seg / A.segs;
        I.c0 = (I.seg - I.b * A.segs) * A.seg_chunks;
        I.c1 = I.c0 + A.seg_chunks < chunks ? I.c0 + A.seg_chunks : chunks;
        I.ch = I.c0 > 0 ? I.c0 - 1 : I.c0;
    };
    auto item_active = [&](const Item &I) -> bool {   // prologue item: only the frame in front of the segment, nothing stored
        const int64_t t = I.ch * NTM + team;
        return t < A.Tf && (I.ch >= I.c0 || team == NTM - 1);
    };
    // the frame in flight
    constexpr int OT = O > 0 ? O : 1;
    constexpr bool LATE3 = PF;   // post filter: rounds 1-3 (12 registers) are requested when the frame is finished — with them in flight too the kernel needs 132 registers, four more than four waves per SIMD have
    f32x4 rv[3], rx01, rcf[OT], rxt[OT];
    float2 r0[8];
    float rgv = 1.f;
    auto issue = [&](const Item &I) {
        if (!item_active(I)) return;
        const int64_t t = I.ch * NTM + team;
        const float2 *Xr = A.spec + (I.b * A.Tf + t) * A.spec_stride;
        int lr = lane;
        DFX_OPAQUE(lr);
        DFX_ASSUME(lr >= 0 && lr < DFX_DSP_TEAM);
        if constexpr (O == 0) {
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int k = lr + u * DFX_DSP_TEAM;
                r0[u] = Xr[k <= M ? k : lr];
            }
        } else {
            // ---- two neighbouring bins per lane, 16-byte accesses.  Round 0 = bins 2 l, 2 l + 1 < 128: the deep-filter bins (k < nb_df <= 128, an even
            // count): taps n read frame t + n - toff; the rest of the round and rounds 1-3: band gains only.
            const f32x4 *Xr4 = reinterpret_cast<const f32x4 *>(Xr);
            const float2 *Cr = A.coefs + I.b * A.cs_b + t * A.cs_t;
            if (A.gains) rgv = A.gains[(I.b * A.Tf + t) * A.nb + (lr < A.nb ? lr : 0)];
            if constexpr (!LATE3) {
#pragma unroll
                for (int u = 1; u < 4; ++u) {
                    const int k = 2 * lr + 128 * u;
                    rv[u - 1] = Xr4[k <= M ? lr + 64 * u : lr];
                }
            }
            const int k = 2 * lr;
            const int kd = k < A.nbdf ? k : 0;   // lanes behind the deep-filter bins load column 0 and drop it (a predicated load: zeroed destination + save / narrow / restore of exec)
            rx01 = Xr4[lr];
            // Every tap is loaded by every lane, without a predicate.  Taps whose frame lies outside the clip contribute nothing: their row index is
            // clamped here and their COEFFICIENT zeroed when the frame is finished, under a wave-uniform branch that only the first / last frames of a
            // clip take.  Gains only (no coefficients): the loads read the frame's own row.
            const bool edge = t - toff < 0 || t + (O - 1) - toff >= A.Tf;
            const bool anydf = A.nbdf > 0;
            const float2 *Xd = Xr + kd, *Cd = anydf ? Cr + kd : Xd;
            const int64_t csn = anydf ? A.cs_n : 0;
            const int sstr = (int)A.spec_stride;
#pragma unroll
            for (int n = 0; n < O; ++n) {
                const int64_t tt = t + n - toff;
                const int dr = edge ? (int)((tt < 0 ? 0 : (tt >= A.Tf ? A.Tf - 1 : tt)) - t) : n - toff;
                rcf[n] = *reinterpret_cast<const f32x4 *>(Cd + n * csn);
                rxt[n] = *reinterpret_cast<const f32x4 *>(Xd + dr * sstr);
            }
        }
    };
    // the frame's values -> gains / deep filter / post filter -> its LDS buffer (same expressions, same order as dfx_k_df_apply_rows: same bits)
    auto finish_frame = [&](int64_t b, int64_t t) {
        int lr = lane;
        DFX_OPAQUE(lr);
        DFX_ASSUME(lr >= 0 && lr < DFX_DSP_TEAM);
        if constexpr (O > 0 && LATE3) {
            const f32x4 *Xr4 = reinterpret_cast<const f32x4 *>(A.spec + (b * A.Tf + t) * A.spec_stride);
#pragma unroll
            for (int u = 1; u < 4; ++u) {
                const int k = 2 * lr + 128 * u;
                rv[u - 1] = Xr4[k <= M ? lr + 64 * u : lr];
            }
        }
        if constexpr (O == 0) {
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int k = lr + u * DFX_DSP_TEAM;
                if (k <= M) bufA[k] = r0[u];
            }
        } else {
            auto gain2 = [&](int u, float &g0, float &g1) {   // (every lane takes part in the gathers)
                g0 = g1 = 1.f;
                if (A.gains) {
                    const unsigned q = bandq[u >> 1] >> (16 * (u & 1));
                    g0 = dfx_lane_gather4(rgv, q & 0xffu), g1 = dfx_lane_gather4(rgv, (q >> 8) & 0xffu);
                }
            };
#pragma unroll
            for (int u = 1; u < 4; ++u) {   // rounds 1-3: band gains only
                const int ku = 2 * lr + 128 * u;
                float gu0, gu1;
                gain2(u, gu0, gu1);
                const f32x4 x = rv[u - 1];
                float2 yu0 = make_float2(x[0] * gu0, x[1] * gu0), yu1 = make_float2(x[2] * gu1, x[3] * gu1);
                if (PF) yu0 = dfx_dfa_finish(yu0, make_float2(x[0], x[1]), A.pf_beta, A.atten_lim), yu1 = dfx_dfa_finish(yu1, make_float2(x[2], x[3]), A.pf_beta, A.atten_lim);
                if (ku <= M) *reinterpret_cast<f32x4 *>(bufA + ku) = f32x4{yu0.x, yu0.y, yu1.x, yu1.y};   // (ku = M: the slot behind the Nyquist bin takes the row's pad bin; nobody reads it)
            }
            const int k = 2 * lr;
            const bool df = k < A.nbdf;
            const f32x4 x01 = rx01;
            if (t - toff < 0 || t + (O - 1) - toff >= A.Tf) {
#pragma unroll
                for (int n = 0; n < O; ++n) {
                    const int64_t tt = t + n - toff;
                    if (tt < 0 || tt >= A.Tf) rcf[n] = f32x4{0.f, 0.f, 0.f, 0.f};
                }
            }
            float g0, g1;
            gain2(0, g0, g1);
            float2 y0, y1;
            if (df) {
                float re0 = 0.f, im0 = 0.f, re1 = 0.f, im1 = 0.f;
#pragma unroll
                for (int n = 0; n < O; ++n) {
                    re0 += rxt[n][0] * rcf[n][0] - rxt[n][1] * rcf[n][1];
                    im0 += rxt[n][0] * rcf[n][1] + rxt[n][1] * rcf[n][0];
                    re1 += rxt[n][2] * rcf[n][2] - rxt[n][3] * rcf[n][3];
                    im1 += rxt[n][2] * rcf[n][3] + rxt[n][3] * rcf[n][2];
                }
                y0 = make_float2(re0, im0), y1 = make_float2(re1, im1);
            } else {
                y0 = make_float2(x01[0] * g0, x01[1] * g0), y1 = make_float2(x01[2] * g1, x01[3] * g1);
            }
            if (PF) y0 = dfx_dfa_finish(y0, make_float2(x01[0], x01[1]), A.pf_beta, A.atten_lim), y1 = dfx_dfa_finish(y1, make_float2(x01[2], x01[3]), A.pf_beta, A.atten_lim);
            *reinterpret_cast<f32x4 *>(bufA + k) = f32x4{y0.x, y0.y, y1.x, y1.y};
        }
    };
    Item cur;
    cur.seg = blockIdx.x;
    bool have = cur.seg < nseg;
    if (have) {
        start_seg(cur);
        issue(cur);
    }
    while (have) {
    const int64_t b = cur.b;
    const bool pro = cur.ch < cur.c0;
    const int64_t t0 = cur.ch * NTM, t = t0 + team;
    const bool active = item_active(cur);
    if (active) finish_frame(b, t);
    Item nxt = cur;
    bool have_next = true;
    if (++nxt.ch >= nxt.c1) {
        nxt.seg += gridDim.x;
        have_next = nxt.seg < nseg;
        if (have_next) start_seg(nxt);
    }
    if (have_next) issue(nxt);
    DFX_WAVE_SYNC();
    // E' = X[k] + conj(X[M-k]) ; O' = conj(w^k) * (X[k] - conj(X[M-k])) ; Z[k] = E' + i*O'   (in place on the pairs (k, M-k), see dfx_k_synthesis).
    // The two bins of a pair share E' and O' up to signs, and w^(M-k) = -conj(w^k): with o = (X[k] - conj(X[M-k])) * conj(w^k),
    // Z[k] = (er - o.y, ei + o.x) and Z[M-k] = (er + o.y, -ei + o.x) — 12 operations per pair instead of 20 (round 6; the sums are the same, only
    // the partner's factor is no longer a table entry of its own)
    if (active) {
        // pairs (k, M - k), k = lane + 64 i: i < 3 always exist (k <= 191), i = 3 for the lanes with k <= M / 2 = 240 (the others read pair 0 and drop
        // it).  Only k = 0 (lane 0 of i = 0) is special: C2R ignores imag(DC) and imag(Nyquist), and its partner slot M is not part of the transform's
        // input — so every pair stores both bins, partner first: lane 0 then leaves a value nobody reads in slot M, and for k = M / 2 = M - k the
        // lane's own second store (LDS operations of a wave complete in order) is the one that stays.  (Round 6: was four predicated iterations
        // with a k == 0 test and a partner test each.)
        constexpr int NPR = (M / 2 + DFX_DSP_TEAM) / DFX_DSP_TEAM;
        float2 xa[NPR], xb[NPR], wa[NPR];
        int lp = lane;
        DFX_OPAQUE(lp);
        DFX_ASSUME(lp >= 0 && lp < DFX_DSP_TEAM);
        const bool last = lp + (NPR - 1) * DFX_DSP_TEAM <= M / 2;
#pragma unroll
        for (int i = 0; i < NPR; ++i) {
            const int k = (i < NPR - 1 || last) ? lp + i * DFX_DSP_TEAM : 0, kc = M - k;
            xa[i] = bufA[k], xb[i] = bufA[kc];
            wa[i] = tp[k];
        }
        if (lp == 0) xa[0].y = 0.f, xb[0].y = 0.f;
#pragma unroll
        for (int i = 0; i < NPR; ++i) {
            const int k = lp + i * DFX_DSP_TEAM, kc = M - k;
            const float2 xk = xa[i], xm = xb[i];
            const float er = xk.x + xm.x, ei = xk.y - xm.y;
            const float2 o = dfx_cmul(make_float2(xk.x - xm.x, xk.y + xm.y), make_float2(wa[i].x, -wa[i].y));
            if (i < NPR - 1 || last) {
                bufA[kc] = make_float2(er + o.y, o.x - ei);
                bufA[k] = make_float2(er - o.y, ei + o.x);
            }
        }
    }
    DFX_WAVE_SYNC();
    if constexpr (MF) dfx_fft480_mfma(bufA, mfr, a3s, lane, active);
    else dfx_fft480_t<+1>(bufA, t1, t2, lane, active);
    __syncthreads();
    // apply_window_in_place (lib.rs:406: the interleaved (re, im) pairs of z ARE the time samples) and the overlap-add in one pass over the LDS:
    // output frame tf = t0 + j gets the windowed second half of frame tf - 1 (the previous wave's buffer, or the carry, which is stored windowed) +
    // the windowed first half of frame tf; the second half of the chunk's last frame becomes the next item's carry.  (Round 6: the window was a
    // pass of its own over every frame buffer — 8 LDS reads + 4 writes of 16 bytes per lane and frame; the products are the same roundings:
    // __fmul_rn / __fadd_rn, never contracted.)
    constexpr int HQ = HOP / 4;
    const float *cin = carry + par * HOP;
    float *cout = carry + (par ^ 1) * HOP;
    auto wmul = [](const f32x4 x, const f32x4 w) -> f32x4 { return f32x4{__fmul_rn(x[0], w[0]), __fmul_rn(x[1], w[1]), __fmul_rn(x[2], w[2]), __fmul_rn(x[3], w[3])}; };
    // Wave j stores output frame t0 + j (round 6: was a flat loop over the chunk's 960 float4 with a division per item): frame, row and the source of the
    // overlap are wave-uniform, a lane's two float4 (lane, lane + 64 < 120) differ by a constant.
    if (!pro) {
        const int64_t tf = t0 + team;
        if (tf < A.Tf) {
            const float *fr = reinterpret_cast<const float *>(bufA);
            int lq = lane;
            DFX_OPAQUE(lq);   // (keeps the lane's share of the addresses out of registers held across the chunk loop)
            DFX_ASSUME(lq >= 0 && lq < DFX_DSP_TEAM);
#pragma unroll
            for (int it = 0; it < (HQ + DFX_DSP_TEAM - 1) / DFX_DSP_TEAM; ++it) {
                const int i = (lq + it * DFX_DSP_TEAM) << 2;
                if (i < HOP) {
                    const f32x4 cur = wmul(*reinterpret_cast<const f32x4 *>(fr + i), *reinterpret_cast<const f32x4 *>(win + i));
                    f32x4 v = cur;
                    if (tf > 0) {
                        const f32x4 old = team > 0 ? wmul(*reinterpret_cast<const f32x4 *>(fr - 2 * BUF + HOP + i), *reinterpret_cast<const f32x4 *>(win + HOP + i))
                                                   : *reinterpret_cast<const f32x4 *>(cin + i);
                        v = f32x4{__fadd_rn(cur[0], old[0]), __fadd_rn(cur[1], old[1]), __fadd_rn(cur[2], old[2]), __fadd_rn(cur[3], old[3])};
                    }
                    dfx_store_out4<I16>(A.out, b * A.out_stride, tf * HOP + i - A.out_skip, A.out_len, v);
                }
            }
        }
    }
    if (threadIdx.x < HQ && t0 + NTM - 1 < A.Tf) {
        const float *fr = reinterpret_cast<const float *>(bufs + (size_t)(NTM - 1) * BUF);
        *reinterpret_cast<f32x4 *>(cout + 4 * threadIdx.x) = wmul(*reinterpret_cast<const f32x4 *>(fr + HOP + 4 * threadIdx.x), *reinterpret_cast<const f32x4 *>(win + HOP + 4 * threadIdx.x));
    }
    par ^= 1;
    __syncthreads();  // the frame buffers are rewritten by the next item, which also reads the carry just stored
    cur = nxt;
    have = have_next;
    }
}

