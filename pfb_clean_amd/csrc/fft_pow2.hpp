// fft_pow2.hpp -- register-resident power-of-two complex FFT shared by a group of
// TPB = N/E threads (wave64-agnostic: groups may be a fraction of a wave or several).
//
// Thread t of the group holds element t + TPB*j in register j (j < E) on entry AND on
// exit (natural order both ways -> coalesced global access with no reordering pass).
// The transform is a Stockham autosort: radix-min(E, remaining) passes done entirely
// in registers, with one LDS exchange between consecutive passes:
//     write  lds[(i-k) R + k + s P]   (i = t + TPB q butterfly, k = i mod P, s < R)
//     read   reg j <- lds[t + TPB j]
// LDS rows are padded by one element every 16 (index + index/16) which makes both the
// strided scatter and the unit-stride gather bank-conflict free for 8- and 16-byte
// elements (guide: ds_write_b64 / ds_read_b64 lane groups).
//
// Twiddles come from a per-plan table with, for every pass s >= 1 and m in {0,1,2,3},
//     ptw[off_s + m P_s + k] = exp(-2 pi i 2^m k / (P_s R_s)),   k < P_s
// so each butterfly does 4 coalesced loads (w, w^2, w^4, w^8) and forms the other
// powers with <= 3 chained complex multiplies (error <= ~5 ulp, measured 2.4).
#pragma once
#include "common.hpp"
#include <type_traits>

namespace pfb {

constexpr int ilog2(int n) { return n <= 1 ? 0 : 1 + ilog2(n >> 1); }
constexpr int imin(int a, int b) { return a < b ? a : b; }

// largest radix done in registers: min(E, 16); a thread with E > 16 elements runs E/16
// radix-16 butterflies per pass
constexpr int rmax_of(int E) { return E < 16 ? E : 16; }

// radix of the pass that starts with P elements already combined
template <int N, int E, int P> struct PassRadix { static constexpr int R = imin(rmax_of(E), N / P); };

// offset (in elements) of pass-with-product-P's twiddle block inside the ptw table
template <int N, int E, int P>
struct PtwOffset {
    // passes before this one: products 1 (no twiddles), E, E^2, ...
    static constexpr int RM = rmax_of(E);
    static constexpr int prev = P / RM;         // product at the previous pass (P = prev * RM)
    static constexpr int value = (P <= RM) ? 0 : PtwOffset<N, E, (P / RM < 1 ? 1 : P / RM)>::value + 4 * prev;
};
template <int N, int E> struct PtwOffset<N, E, 1> { static constexpr int value = 0; };

template <int N, int E>
constexpr int ptw_total() {
    int tot = 0;
    for (int P = rmax_of(E); P < N; P *= rmax_of(E)) tot += 4 * P;
    return tot;
}

// host: fill the table (long double accuracy, rounded once)
template <typename T, int N, int E>
void fill_ptw(cplx<T>* out) {
    const long double two_pi = 6.283185307179586476925286766559005768L;
    int off = 0;
    for (int P = rmax_of(E); P < N; P *= rmax_of(E)) {
        const int R = imin(rmax_of(E), N / P);
        for (int m = 0; m < 4; ++m)
            for (int k = 0; k < P; ++k) {
                long double a = two_pi * (long double)((long long)(1 << m) * k) / (long double)((long long)P * R);
                out[off + m * P + k] = cplx<T>((T)cosl(a), (T)(-sinl(a)));
            }
        off += 4 * P;
    }
}

// compact table: w only (the m = 0 rows of fill_ptw), offsets off_s / 4
template <typename T, int N, int E>
void fill_ptw_compact(cplx<T>* out) {
    const long double two_pi = 6.283185307179586476925286766559005768L;
    int off = 0;
    for (int P = rmax_of(E); P < N; P *= rmax_of(E)) {
        const int R = imin(rmax_of(E), N / P);
        for (int k = 0; k < P; ++k) {
            long double a = two_pi * (long double)k / (long double)((long long)P * R);
            out[off + k] = cplx<T>((T)cosl(a), (T)(-sinl(a)));
        }
        off += P;
    }
}

// ------------------------------------------------------------- small in-register DFTs
template <typename T, bool INV>
__device__ __forceinline__ cplx<T> rot90(cplx<T> a) {   // multiply by -i (fwd) / +i (inv)
    if constexpr (sizeof(T) == 4) {
        v2f r;          // one packed add with 0: swizzle + half negation in the modifiers
        if constexpr (INV) asm("v_pk_add_f32 %0, %1, 0 op_sel:[1,0] op_sel_hi:[0,0] neg_lo:[1,0]" : "=v"(r) : "v"(a.v));
        else               asm("v_pk_add_f32 %0, %1, 0 op_sel:[1,0] op_sel_hi:[0,0] neg_hi:[1,0]" : "=v"(r) : "v"(a.v));
        return cplx<T>(r);
    } else {
        return INV ? cplx<T>(-a.y, a.x) : cplx<T>(a.y, -a.x);
    }
}

template <typename T, bool INV>
__device__ __forceinline__ void dft2(cplx<T>& a, cplx<T>& b) {
    cplx<T> t = a - b;
    a = a + b;
    b = t;
}

template <typename T, bool INV>
__device__ __forceinline__ void dft4(cplx<T>& a0, cplx<T>& a1, cplx<T>& a2, cplx<T>& a3) {
    const cplx<T> t0 = a0 + a2, t1 = a0 - a2, t2 = a1 + a3, d = a1 - a3;
    a0 = t0 + t2;
    a1 = addrot<INV>(t1, d);         // t1 + rot(d)   (one packed add with swizzle for fp32)
    a2 = t0 - t2;
    a3 = addrot<!INV>(t1, d);        // t1 - rot(d)
}

// multiply by exp(-+ 2 pi i m / 16) with compile-time m
template <typename T, bool INV, int m>
__device__ __forceinline__ cplx<T> mul_w16(cplx<T> a) {
    constexpr T c = T(0.92387953251128675612818318939678828682L);
    constexpr T s = T(0.38268343236508977172845998403039886676L);
    constexpr T h = T(0.70710678118654752440084436210484903928L);
    constexpr int mm = m & 15;
    if constexpr (mm == 0) return a;
    else if constexpr (mm == 4) return rot90<T, INV>(a);
    else if constexpr (mm == 8) return cplx<T>(-a.x, -a.y);
    else if constexpr (mm == 12) return rot90<T, !INV>(a);
    else if constexpr (mm == 2)  return h * (INV ? mul_1pi(a) : mul_1mi(a));
    else if constexpr (mm == 14) return h * (INV ? mul_1mi(a) : mul_1pi(a));
    else if constexpr (mm == 6)  return (-h) * (INV ? mul_1mi(a) : mul_1pi(a));
    else if constexpr (mm == 10) return (-h) * (INV ? mul_1pi(a) : mul_1mi(a));
    else {
        // w = (wr, -wi) forward, (wr, +wi) inverse
        constexpr T wr = (mm == 1) ? c : (mm == 3) ? s : (mm == 5) ? -s : (mm == 7) ? -c : (mm == 9) ? -c
                       : (mm == 11) ? -s : (mm == 13) ? s : c;
        constexpr T wi = (mm == 1) ? s : (mm == 3) ? c : (mm == 5) ? c : (mm == 7) ? s : (mm == 9) ? -s
                       : (mm == 11) ? -c : (mm == 13) ? -c : -s;
        const T wy = INV ? wi : -wi;
        return a * cplx<T>(wr, wy);
    }
}

template <typename T, bool INV, int R> struct Dft;

template <typename T, bool INV> struct Dft<T, INV, 2> {
    __device__ __forceinline__ static void run(cplx<T> (&u)[2]) { dft2<T, INV>(u[0], u[1]); }
};
template <typename T, bool INV> struct Dft<T, INV, 4> {
    __device__ __forceinline__ static void run(cplx<T> (&u)[4]) { dft4<T, INV>(u[0], u[1], u[2], u[3]); }
};
template <typename T, bool INV> struct Dft<T, INV, 8> {
    // n = nl + 2 nh : inner dft4 over nh, twiddle w8^(nl k1), outer dft2 over nl -> X[k1 + 4 k2]
    __device__ __forceinline__ static void run(cplx<T> (&u)[8]) {
        dft4<T, INV>(u[0], u[2], u[4], u[6]);
        dft4<T, INV>(u[1], u[3], u[5], u[7]);
        u[3] = mul_w16<T, INV, 2>(u[3]);
        u[7] = mul_w16<T, INV, 6>(u[7]);
        // now b[nl][k1] sits in u[nl + 2 k1]; the w8^2 = -+i rotation of u[5] rides on the adds
        cplx<T> x[8];
#pragma unroll
        for (int k1 = 0; k1 < 4; ++k1) {
            const cplx<T> p = u[2 * k1], q = u[2 * k1 + 1];
            if (k1 == 2) {
                x[k1] = addrot<INV>(p, q);
                x[k1 + 4] = addrot<!INV>(p, q);
            } else {
                x[k1] = p + q;
                x[k1 + 4] = p - q;
            }
        }
#pragma unroll
        for (int k = 0; k < 8; ++k) u[k] = x[k];
    }
};
template <typename T, bool INV> struct Dft<T, INV, 16> {
    // n = nl + 4 nh : inner dft4 over nh (b[nl][k1] in u[nl + 4 k1]), twiddle w16^(nl k1),
    // outer dft4 over nl -> X[k1 + 4 k2]
    __device__ __forceinline__ static void run(cplx<T> (&u)[16]) {
        dft4<T, INV>(u[0], u[4], u[8], u[12]);
        dft4<T, INV>(u[1], u[5], u[9], u[13]);
        dft4<T, INV>(u[2], u[6], u[10], u[14]);
        dft4<T, INV>(u[3], u[7], u[11], u[15]);
        u[5] = mul_w16<T, INV, 1>(u[5]);   u[6] = mul_w16<T, INV, 2>(u[6]);   u[7] = mul_w16<T, INV, 3>(u[7]);
        u[9] = mul_w16<T, INV, 2>(u[9]);   u[10] = mul_w16<T, INV, 4>(u[10]); u[11] = mul_w16<T, INV, 6>(u[11]);
        u[13] = mul_w16<T, INV, 3>(u[13]); u[14] = mul_w16<T, INV, 6>(u[14]); u[15] = mul_w16<T, INV, 9>(u[15]);
        // outer: for each k1 combine u[0 + 4k1], u[1 + 4k1], u[2 + 4k1], u[3 + 4k1] -> X[k1 + 4 k2]
        dft4<T, INV>(u[0], u[1], u[2], u[3]);       // k1 = 0 -> X[0], X[4], X[8], X[12]
        dft4<T, INV>(u[4], u[5], u[6], u[7]);       // k1 = 1 -> X[1], X[5], X[9], X[13]
        dft4<T, INV>(u[8], u[9], u[10], u[11]);     // k1 = 2 -> X[2], X[6], X[10], X[14]
        dft4<T, INV>(u[12], u[13], u[14], u[15]);   // k1 = 3 -> X[3], X[7], X[11], X[15]
        // u[4 k1 + k2] = X[k1 + 4 k2]  -> transpose the 4x4 index to natural order
        cplx<T> t;
        t = u[1];  u[1] = u[4];   u[4] = t;
        t = u[2];  u[2] = u[8];   u[8] = t;
        t = u[3];  u[3] = u[12];  u[12] = t;
        t = u[6];  u[6] = u[9];   u[9] = t;
        t = u[7];  u[7] = u[13];  u[13] = t;
        t = u[11]; u[11] = u[14]; u[14] = t;
    }
};

// ------------------------------------------------------------------- the group FFT
// WAVE: the TPB threads of a group sit inside ONE wavefront (TPB <= 64): the LDS exchanges
// then need no s_barrier at all -- a wave's DS instructions execute in order, so its reads
// see its own earlier writes; only the compiler must be kept from reordering them.
// DBOFF > 0: a second LDS buffer sits DBOFF elements after the first and the exchanges
// alternate between them (first exchange -> alternate buffer), which removes the barrier
// in front of every write: the buffer being written was last read two barriers ago.
// SQTW: `ptw` is the COMPACT table (w only: ptw_c[off_s/4 + k]) and w^2, w^4, w^8 are
// formed by squaring.  The compact table is small enough (sum of P_s elements) to be copied
// into LDS by the kernel, which takes every global load out of the passes: s_waitcnt vmcnt
// is in-order, so a twiddle load inside a pass would force every older, deliberately
// early-issued load (prefetch of the next operand) to complete first.
// SEQX: the NV transforms of runN share ONE LDS buffer and exchange one after the other
// (twice the barriers per pass, NV times less LDS).
// Pass hook.  A persistent kernel that wants memory traffic in flight WHILE a transform runs hands runN / run a
// callable that is invoked once at the top of every pass with std::integral_constant<int, pass index>: it issues a
// compile-time slice of the kernel's prefetch loads there.  Why: a wave that issues more vector-memory instructions
// than the CU's memory pipeline has room for simply STALLS AT ISSUE until older requests have returned
// (profiles/r02_a_phase_stamps: "issue x, r, next y + barrier" = 7.6 us of a 22.6 us row-inverse trip, with every
// wave of the one resident workgroup stalled together and nothing computing).  A few loads per pass keep the
// pipeline full without parking all the waves at once.
struct NoPassHook {
    template <typename K> __device__ __forceinline__ void operator()(K) const {}
};
template <int K> using PassIdx = std::integral_constant<int, K>;

template <typename T, int N, int E, bool WAVE = false, int DBOFF = 0, bool SQTW = false, bool SEQX = false>
struct RegFft {
    static_assert((N & (N - 1)) == 0 && (E & (E - 1)) == 0 && N >= E, "power-of-two sizes");
    static_assert(!WAVE || N / E <= 64, "wave-local exchange needs the group inside one wave");
    __device__ __forceinline__ static void sync() {
        if constexpr (WAVE) __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        else __syncthreads();
    }
    static constexpr int TPB = N / E;
    static constexpr int RM = rmax_of(E);
    static constexpr int LDS_ELEMS = N + N / (RM < (128 / (int)sizeof(cplx<T>)) ? RM : (128 / (int)sizeof(cplx<T>)));   // room for every padding below
    static constexpr int PTW = ptw_total<N, E>();
    static constexpr int PTWC = ptw_total<N, E>() / 4;          // compact (w only) table size

    __device__ __forceinline__ static int pad(int i) { return i + (i >> 4); }
    // pad(b + c) == pad(b) + cpad(c) whenever (b & 15) + (c & 15) < 16, which holds for
    // every (base, constant) pair used below; it lets the 16 LDS accesses of a thread
    // share ONE address register with compile-time immediates.
    static constexpr int cpad(int c) { return c + (c >> 4); }

    // Padding of the EXCHANGE after the pass with predecessor product P (radix R), chosen per
    // pass so that the scattered writes are bank-conflict free (the reads are contiguous).
    // The LDS behaves as 32 banks x 4 bytes = 128 bytes per clock (SQ_LDS_BANK_CONFLICT on
    // stride-2 fp32 reads: exactly 50 %), i.e. SLOTS = 16 complex64 / 8 complex128 elements
    // served per clock, taken by SLOTS consecutive lanes:
    //   P = 1   lane i writes element i*R (+s): the SLOTS lanes of a group would share 2 (1)
    //           slots -> shift by one element per SLOTS elements: c + c/SLOTS
    //   P > 1   P consecutive lanes write P contiguous elements, the next P lanes start P*R
    //           further (a multiple of SLOTS): shift every P*R block by P elements; nothing
    //           needed once P lanes alone fill a clock (P >= SLOTS).
    // (measured on the column kernel: uniform i + i/16 everywhere 35 % of the LDS cycles in
    // conflicts; P-dependent shifts as below but c/32 for P = 1: 15 %)
    // As for pad/cpad above: xpad(b + c) == xpad(b) + cxpad(c) for every (base, constant) used.
    static constexpr int CB = (int)sizeof(cplx<T>);
    static constexpr int SLOTS = 128 / CB;
    template <int P, int R>
    static constexpr int cxpad(int c) {
        if (P == 1) return c + c / SLOTS;
        if (P >= SLOTS) return c;
        return c + (c / (P * R)) * P;
    }
    template <int P, int R>
    __device__ __forceinline__ static int xpad(int c) { return cxpad<P, R>(c); }

    // twiddle + butterflies of the pass whose predecessor product is P
    template <bool INV, int P>
    __device__ __forceinline__ static void butterflies(cplx<T> (&v)[E], int t, const cplx<T>* __restrict__ ptw) {
        constexpr int R = PassRadix<N, E, P>::R;
        constexpr int NB = E / R;                 // butterflies per thread
#pragma unroll
        for (int q = 0; q < NB; ++q) {
            cplx<T> u[R];
#pragma unroll
            for (int r = 0; r < R; ++r) u[r] = v[q + r * NB];
            if constexpr (P > 1) {
                const int k = (t + TPB * q) & (P - 1);
                const cplx<T>* tb = ptw + (SQTW ? PtwOffset<N, E, P>::value / 4 : PtwOffset<N, E, P>::value) + k;
                // w, w^2, w^4, w^8 are loaded (or squared up from w); the other powers are
                // formed right where they are consumed so that few twiddles are live at a time
                cplx<T> w1 = tb[0], w2, w4, w8;
                if constexpr (SQTW) {
                    if constexpr (R > 2) w2 = w1 * w1;
                    if constexpr (R > 4) w4 = w2 * w2;
                    if constexpr (R > 8) w8 = w4 * w4;
                } else {
                    if constexpr (R > 2) w2 = tb[P];
                    if constexpr (R > 4) w4 = tb[2 * P];
                    if constexpr (R > 8) w8 = tb[3 * P];
                }
                // inverse transform: multiply by the conjugates (mulc), no sign flips on w
                auto tw = [](cplx<T> a, cplx<T> w) { return INV ? mulc(a, w) : a * w; };
                u[1] = tw(u[1], w1);
                if constexpr (R > 2) {
                    u[2] = tw(u[2], w2);
                    const cplx<T> w3 = w1 * w2;
                    u[3] = tw(u[3], w3);
                    if constexpr (R > 4) {
                        u[4] = tw(u[4], w4);
                        const cplx<T> w7 = w3 * w4;
                        u[7] = tw(u[7], w7);
                        if constexpr (R > 8) { u[11] = tw(u[11], w3 * w8); u[15] = tw(u[15], w7 * w8); }
                        const cplx<T> w5 = w1 * w4;
                        u[5] = tw(u[5], w5);
                        if constexpr (R > 8) u[13] = tw(u[13], w5 * w8);
                        const cplx<T> w6 = w2 * w4;
                        u[6] = tw(u[6], w6);
                        if constexpr (R > 8) {
                            u[14] = tw(u[14], w6 * w8);
                            u[8] = tw(u[8], w8);
                            u[9] = tw(u[9], w1 * w8);
                            u[10] = tw(u[10], w2 * w8);
                            u[12] = tw(u[12], w4 * w8);
                        }
                    }
                }
            }
            Dft<T, INV, R>::run(u);
#pragma unroll
            for (int r = 0; r < R; ++r) v[q + r * NB] = u[r];
        }
    }

    // NV independent transforms advance together: same barriers, NV LDS buffers
    // (lds + n * LDS_ELEMS), NV * E values in registers.  NV = 1 for the row kernels,
    // NV = 2 for the column kernel (two frequency columns per 16-byte access).
    template <bool INV, int P, int NV, int XCH = 0, int K = 0, typename Hook = NoPassHook>
    __device__ __forceinline__ static void pass(cplx<T> (&v)[NV][E], cplx<T>* lds_in, int t,
                                                const cplx<T>* __restrict__ ptw, const Hook& hook = Hook()) {
        constexpr int R = PassRadix<N, E, P>::R;
        constexpr int NB = E / R;
        hook(PassIdx<K>{});
#pragma unroll
        for (int n = 0; n < NV; ++n) butterflies<INV, P>(v[n], t, ptw);
        if constexpr (P * R < N) {
            cplx<T>* lds = (DBOFF > 0 && (XCH & 1) == 0) ? lds_in + DBOFF : lds_in;
            if constexpr (SEQX) {
#pragma unroll
                for (int n = 0; n < NV; ++n) {
                    sync();                        // previous readers of `lds` are done
#pragma unroll
                    for (int q = 0; q < NB; ++q) {
                        const int i = t + TPB * q;
                        const int k = i & (P - 1);
                        cplx<T>* wp = lds + xpad<P, R>((i - k) * R + k);
#pragma unroll
                        for (int s = 0; s < R; ++s) wp[cxpad<P, R>(s * P)] = v[n][q + s * NB];
                    }
                    sync();
                    const cplx<T>* rp = lds + xpad<P, R>(t);
#pragma unroll
                    for (int j = 0; j < E; ++j) v[n][j] = rp[cxpad<P, R>(TPB * j)];
                }
            } else {
            if constexpr (DBOFF == 0) sync();      // previous readers of `lds` are done
#pragma unroll
            for (int q = 0; q < NB; ++q) {
                const int i = t + TPB * q;
                const int k = i & (P - 1);
                cplx<T>* wp = lds + xpad<P, R>((i - k) * R + k);
#pragma unroll
                for (int n = 0; n < NV; ++n) {
#pragma unroll
                    for (int s = 0; s < R; ++s) wp[n * LDS_ELEMS + cxpad<P, R>(s * P)] = v[n][q + s * NB];
                }
            }
            sync();
            const cplx<T>* rp = lds + xpad<P, R>(t);
#pragma unroll
            for (int n = 0; n < NV; ++n) {
#pragma unroll
                for (int j = 0; j < E; ++j) v[n][j] = rp[n * LDS_ELEMS + cxpad<P, R>(TPB * j)];
            }
            }
            pass<INV, P * R, NV, XCH + 1, K + 1, Hook>(v, lds_in, t, ptw, hook);
        }
    }

    // All threads of the WORKGROUP must call (contains __syncthreads); `lds` is this
    // group's private buffer of NV * LDS_ELEMS elements.
    // X0: parity of the first exchange (DBOFF > 0 only).  A transform with an ODD number of exchanges
    // ends on the buffer it started on, so back-to-back transforms without a barrier in between must
    // alternate X0 (0, 1, 0, ...) to keep "the buffer being written was last read two barriers ago".
    template <bool INV, int NV, int X0 = 0, typename Hook = NoPassHook>
    __device__ __forceinline__ static void runN(cplx<T> (&v)[NV][E], cplx<T>* lds, int t,
                                                const cplx<T>* __restrict__ ptw, const Hook& hook = Hook()) {
        pass<INV, 1, NV, X0, 0, Hook>(v, lds, t, ptw, hook);
    }
    template <int P = 1> static constexpr int nxch() {       // LDS exchanges per transform
        constexpr int R = PassRadix<N, E, P>::R;
        if constexpr (P * R < N) return 1 + nxch<P * R>();
        else return 0;
    }
    static constexpr int NPASS = nxch<1>() + 1;               // passes per transform (= hook calls)
    template <bool INV, typename Hook = NoPassHook>
    __device__ __forceinline__ static void run(cplx<T> (&v)[E], cplx<T>* lds, int t,
                                               const cplx<T>* __restrict__ ptw, const Hook& hook = Hook()) {
        pass<INV, 1, 1, 0, 0, Hook>(reinterpret_cast<cplx<T>(&)[1][E]>(v), lds, t, ptw, hook);
    }
};

}  // namespace pfb
