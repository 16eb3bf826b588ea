// Device code of the serial link of a horizon step (glue_body), shared by its two hosts: k_glue (glue.hip) and the
// fused head k_mm_prep<DT, true> (prep.hip).  Internal; gfx950 only.
#pragma once
#include "mm_device.h"

namespace pilco {

// ------------------------------------------------------------------ glue
// The serial link of a horizon step: everything it needs is pulled into LDS with ONE batch of loads (all requests in
// flight before the first wait), then (pack ->) assemble -> propagate -> controller -> joint Gaussian.
// Two hosts run the same code (glue_body):
//   * k_glue: one workgroup (+ an optional second one evaluating the reward of the pre-propagation state);
//   * k_mm_prep<DT, true> ("fused head", prep.hip): EVERY workgroup of the next step's prep launch runs the link
//     redundantly on its own CU and goes straight on to its prep work with the new joint Gaussian in LDS -- no launch
//     boundary and no global round trip between the two; only workgroup (0,0) (`writer`) stores the results.
struct GlueLds {
    double* mx;   // [nm]     current state mean
    double* sx;   // [nm*nm]  current state covariance
    double* mu;   // [nm]
    double* su;   // [nm*nm]
    double* cxu;  // [nm*nm]
    double* t1;   // [nm*nm]
    double* t2;   // [nm*nm]
    double* s1;   // [nm*nm]  s1 = [s_x, s_x c_xu] of the previous joint
    double* jm;   // [nm]     joint mean handed to the dynamics GP
    double* js;   // [nm*nm]  joint covariance
    double* seg;  // [SEG]    this rank's packed results
    double* mp;   // [EL*NCH*(1+D)] mean partials
    double* misc; // [256]: [1..) cdiag, [64..) / [96..) temporaries, [128..) policy bias b, [160..) max_action (batched load)
    double* pol;  // scratch of the inline RbfController evaluation (GlueArgs::pol_lds doubles), behind mp
    double* xm;   // fused head: where the HOST workgroup's next phase wants the joint mean [D] and covariance [D][D] (its own
    double* xs;   // LDS region): write_joint stores them there too, and that phase starts without a copy and a barrier of its own
    int nm;       // max(E, D): leading dimension of the square buffers
    int o_sx, o_s1, o_js, o_seg, o_mp, o_misc;   // offsets (doubles) of sx, s1, js, seg, mp from mx, for multi_load
};

__device__ __forceinline__ void glue_lds_carve(const GlueArgs& g, double* sm, GlueLds& L) {
    const int E = g.E, D = g.D;
    const int nm = E > D ? E : D;
    int seg_n = (g.flags & (GF_PACK | GF_ASSEMBLE)) ? g.wk.SEG * ((g.flags & GF_PACK) ? 1 : g.wk.nranks) : 0;
    if (g.flags & GF_RBF_POST) seg_n = g.pwk.SEG;
    L.mx = sm;
    L.sx = L.mx + nm;
    L.mu = L.sx + nm * nm;
    L.su = L.mu + nm;
    L.cxu = L.su + nm * nm;
    L.t1 = L.cxu + nm * nm;
    L.t2 = L.t1 + nm * nm;
    L.s1 = L.t2 + nm * nm;
    L.jm = L.s1 + nm * nm;
    L.js = L.jm + nm;
    L.misc = L.js + nm * nm;
    L.seg = L.misc + 256;
    L.mp = L.seg + seg_n;
    {
        const int mp_n = (g.flags & GF_RBF_POST) ? g.pwk.EL * g.pwk.NCHM * (1 + E) : ((g.flags & GF_PACK) ? g.wk.EL * g.wk.NCHM * (1 + D) : 0);
        const int rew_n = (int)reward_lds_doubles(E);
        const int tail = seg_n + mp_n;
        L.pol = L.seg + (tail > rew_n ? tail : rew_n);   // (matches glue_lds_doubles: the tail region is max(seg + mp, reward scratch))
    }
    L.xm = nullptr;
    L.xs = nullptr;
    L.nm = nm;
    L.o_sx = nm;
    L.o_s1 = 2 * nm + 5 * nm * nm;
    L.o_js = 3 * nm + 6 * nm * nm;
    L.o_seg = 3 * nm + 7 * nm * nm + 256;
    L.o_misc = 3 * nm + 7 * nm * nm;
    L.o_mp = L.o_seg + seg_n;
}

// Batched global -> LDS copies: up to six segments are treated as one index space; every thread requests all its
// elements (MAXV per round) before the first one is consumed, so the whole batch costs ONE memory round trip instead
// of one per segment (the first version copied segment after segment: ~1 us each on this serial path).
// (Destinations are offsets from one LDS base, not pointers: an aggregate of LDS pointers makes this compiler fold the
// LDS -> generic cast of the region's first address into an illegal instruction.)
struct LoadSeg {
    int dst;            // offset (doubles) from the LDS base passed to multi_load
    const double* src;  // global
    int n;
};
template <int NSEG, int MAXV>
__device__ __forceinline__ void multi_load(double* lds, const LoadSeg (&sg)[NSEG]) {
    int off[NSEG + 1];
    off[0] = 0;
#pragma unroll
    for (int i = 0; i < NSEG; ++i) off[i + 1] = off[i] + sg[i].n;
    const int total = off[NSEG];
    const int nthr = (int)blockDim.x;
    for (int base = 0; base < total; base += MAXV * nthr) {
        double v[MAXV];
#pragma unroll
        for (int k = 0; k < MAXV; ++k) {
            const int e = base + k * nthr + (int)threadIdx.x;
            v[k] = 0.0;
            if (e < total) {
#pragma unroll
                for (int i = 0; i < NSEG; ++i)
                    if (e >= off[i] && e < off[i + 1]) v[k] = sg[i].src[e - off[i]];
            }
        }
#pragma unroll
        for (int k = 0; k < MAXV; ++k) {
            const int e = base + k * nthr + (int)threadIdx.x;
            if (e < total) {
#pragma unroll
                for (int i = 0; i < NSEG; ++i)
                    if (e >= off[i] && e < off[i + 1]) lds[sg[i].dst + (e - off[i])] = v[k];
            }
        }
    }
}

// squash_sin on (mu[U], su[U][U]) in place; cdiag[u] = e_u exp(-s_uu/2) cos(m_u)   (controllers.py:13-36).
// The 5 U^2 + 3 U transcendental evaluations of the reference's formulas (exp(lq), exp(lq +- s), cos(m_u -+ m_v) per
// element; exp, cos, sin per control) are independent: each goes to its own thread (one library call deep instead of a
// chain of five on this serial path), then one combining phase.  Scratch: t1 .. js (contiguous, dead at this point).
// eval_only (sc_alt: scratch that must outlive the call; one round, 5 (U^2 + U) slots): stop behind the evaluations' barrier and
// leave them in sc_alt -- the caller combines them where it needs them (write_joint_lin_squash below).
__device__ __forceinline__ void squash_inplace(const GlueLds& L, int U, const double* maxact, double* cdiag, double* sc_alt = nullptr, bool eval_only = false) {
    const int t = threadIdx.x;
    const int nm = L.nm, nm_sq = nm * nm;
    // ONE inlined copy of each library function (exp, cos, sin), whatever element a thread evaluates: the argument and
    // the function are selected first.  (The first version had a copy per branch and a second, sequential code path for
    // large U -- 14 inlined transcendental bodies, ~20 KB of a link whose instruction stream is on the step's critical
    // path)  The 5 U^2 + 3 U evaluations of the reference's formulas (exp(lq), exp(lq +- s),
    // cos(m_u -+ m_v) per element; exp, cos, sin per control) are independent: each goes to its own thread, in rounds when
    // the scratch (t1 .. js: 4 nm^2 + nm values) is smaller than that; then one combining phase per round.
    double* sc = sc_alt ? sc_alt : L.t1;
    const int cap = sc_alt ? 5 * (U * U + U) : ((4 * nm_sq + nm) / 5) * 5;      // whole items per round: an item is 5 slots (a control uses 3 of its 5)
    const int nitems = U * U + U;                    // items 0 .. U^2 - 1: covariance elements, then the U controls
    // the new covariance must not overwrite su while later rounds still read it: it is collected in registers per thread
    // (element e = t + k * blockDim of the U x U matrix; U <= 32, blockDim >= 256: k < 4) and stored after the last round
    double newS[4] = {0.0, 0.0, 0.0, 0.0};
    double newM = 0.0, newC = 0.0;
    for (int i0 = 0; i0 < 5 * nitems; i0 += cap) {
        const int n = (5 * nitems - i0) < cap ? (5 * nitems - i0) : cap;
        // A wave evaluates ONE of the three functions (wave w: function w % 3; a wave whose lanes ask for different functions
        // would run through all three bodies, 1.3 us of this serial path instead of 0.5): the round's slots of function f are
        // dealt over the lanes of the waves that serve f.  Items [qa, qb) of this round: nc covariance elements, then nu controls.
        {
            const int w = t >> 6, f = w % 3, nw = (int)blockDim.x >> 6;
            const int nwf = (nw - f + 2) / 3;                       // waves that serve function f
            const int qa = i0 / 5, qb = (i0 + n) / 5;
            const int nc = (U * U > qa) ? ((U * U < qb ? U * U : qb) - qa) : 0, nu = (qb - qa) - nc;
            const int cnt = (f == 0) ? 3 * nc + nu : (f == 1) ? 2 * nc + nu : nu;
            for (int e = (w / 3) * 64 + (t & 63); e < cnt; e += nwf * 64) {
                int q, j;                                           // item and its slot
                if (f == 0) {
                    if (e < 3 * nc) { q = qa + e / 3; j = e - 3 * (e / 3); } else { q = qa + nc + (e - 3 * nc); j = 0; }
                } else if (f == 1) {
                    if (e < 2 * nc) { q = qa + (e >> 1); j = 3 + (e & 1); } else { q = qa + nc + (e - 2 * nc); j = 1; }
                } else {
                    q = qa + nc + e; j = 2;
                }
                double arg;
                if (q < U * U) {
                    int v;
                    const int u = idiv_s(q, U, v);
                    const double lq = -(L.su[u * U + u] + L.su[v * U + v]) / 2.0;
                    const double suv = L.su[q];
                    arg = (j == 0) ? lq : (j == 1) ? lq + suv : (j == 2) ? lq - suv : (j == 3) ? L.mu[u] - L.mu[v] : L.mu[u] + L.mu[v];
                } else {
                    const int u = q - U * U;
                    arg = (j == 0) ? -L.su[u * U + u] / 2.0 : L.mu[u];
                }
                double val;
                if (f == 0) val = exp(arg);                         // (wave-uniform: one body per wave)
                else if (f == 1) val = cos(arg);
                else val = sin(arg);
                sc[5 * (q - qa) + j] = val;
            }
        }
        __syncthreads();
        if (eval_only) return;   // (uniform)
        const int q0 = i0 / 5, q1 = (i0 + n) / 5;    // items [q0, q1) are complete in this round
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int e = t + k * (int)blockDim.x;
            if (e < U * U && e >= q0 && e < q1) {
                int v;
                const int u = idiv_s(e, U, v);
                const double* r = sc + 5 * (e - q0);
                const double val = (r[1] - r[0]) * r[3] - (r[2] - r[0]) * r[4];
                const double eu = maxact ? maxact[u] : 1.0, ev = maxact ? maxact[v] : 1.0;
                newS[k] = eu * ev * val / 2.0;
            }
        }
        if (t < U && U * U + t >= q0 && U * U + t < q1) {
            const double eu = maxact ? maxact[t] : 1.0;
            const double* r = sc + 5 * (U * U + t - q0);
            newC = eu * r[0] * r[1];
            newM = eu * r[0] * r[2];
        }
        if (i0 + cap < 5 * nitems) __syncthreads();   // (another round overwrites the scratch; the stores below touch su / mu / cdiag only)
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int e = t + k * (int)blockDim.x;
        if (e < U * U) L.su[e] = newS[k];
    }
    if (t < U) {
        cdiag[t] = newC;
        L.mu[t] = newM;
    }
    __syncthreads();
}

// joint Gaussian of (x,u) from mx,sx,mu,su,cxu in LDS -> jm, js in LDS (pilco.py:141-144); the writer workgroup also
// stores in_m, in_s, s1 (the next propagate's [s_x, s_x c_xu]) and the tape record
// cdiag != nullptr: c_xu is still to be scaled by the squash's diagonal C (V @ C, controllers.py:35) -- done on the fly here,
// product first as the separate pass rounds it, instead of in a phase of its own (a barrier and an LDS round trip of the link)
__device__ __forceinline__ void write_joint(const GlueArgs& g, const GlueLds& L, bool writer, const double* cdiag = nullptr) {
    const int E = g.E, U = g.U, D = g.D, t = threadIdx.x;
    for (int e = t; e < E * U; e += blockDim.x) {  // sc = s_x c_xu  (E,U)
        int u;
        const int r = idiv_s(e, U, u);
        double acc = 0.0;
        if (cdiag) {
            const double cd = cdiag[u];
            _Pragma("unroll 8") for (int k = 0; k < E; ++k) acc = fma(L.sx[r * E + k], L.cxu[k * U + u] * cd, acc);
        } else {
            _Pragma("unroll 8") for (int k = 0; k < E; ++k) acc = fma(L.sx[r * E + k], L.cxu[k * U + u], acc);
        }
        L.t1[e] = acc;
    }
    if (U > 0) __syncthreads();
    double* s1_dst = g.s1_out ? g.s1_out : g.s1;
    if (t < D) {
        const double v = (t < E) ? L.mx[t] : L.mu[t - E];
        L.jm[t] = v;
        if (L.xm) L.xm[t] = v;
        if (writer) g.wk.in_m[t] = v;
    }
    for (int e = t; e < D * D; e += blockDim.x) {
        int c;
        const int r = idiv_s(e, D, c);
        double v;
        if (r < E && c < E) v = L.sx[r * E + c];
        else if (r < E) v = L.t1[r * U + (c - E)];
        else if (c < E) v = L.t1[c * U + (r - E)];
        else v = L.su[(r - E) * U + (c - E)];
        L.js[e] = v;
        if (L.xs) L.xs[e] = v;
        if (writer) {
            g.wk.in_s[e] = v;
            if (r < E) s1_dst[r * D + c] = v;
            if (g.tape) {
                double* rec = g.tape + (long)g.step * (D + D * D + E * D + E + E * E + D * E);
                rec[D + e] = v;
                if (r < E) rec[D + D * D + r * D + c] = v;
            }
        }
    }
    if (writer && g.tape && t < D) g.tape[(long)g.step * (D + D * D + E * D + E + E * E + D * E) + t] = (t < E) ? L.mx[t] : L.mu[t - E];
    __syncthreads();
}

// LinearController + squash_sin, U small: the joint Gaussian straight from the squash's EVALUATIONS (sq: squash_inplace's slots,
// item (u, v): exp(lq) exp(lq + s) exp(lq - s) cos(m_u - m_v) cos(m_u + m_v); control u: exp(-s_uu / 2) cos m_u sin m_u) and the
// controller's W s_x (L.t1): every thread combines what its own entry needs --
//   mean E + u:  e_u exp(-s_uu/2) sin m_u;   S_uv = e_u e_v ((E+ - E0) cos(m_u - m_v) - (E- - E0) cos(m_u + m_v)) / 2;
//   s_x c_xu C = (W s_x)^T C   (c_xu = W^T: no second E x E x U product),  C_u = e_u exp(-s_uu/2) cos m_u
// -- instead of squash_inplace's combining phase + write_joint's product phase: two barrier intervals and two LDS round trips
// less on the step's serial path (1 us of the link's 4 us joint phase at C2u).  Stores: as write_joint.
__device__ __forceinline__ void write_joint_lin_squash(const GlueArgs& g, const GlueLds& L, bool writer, const double* sq, const double* maxact) {
    const int E = g.E, U = g.U, D = g.D, t = threadIdx.x;
    double* s1_dst = g.s1_out ? g.s1_out : g.s1;
    auto cdiag_of = [&](int u) { const double* r = sq + 5 * (U * U + u); return (maxact ? maxact[u] : 1.0) * r[0] * r[1]; };
    if (t < D) {
        double v;
        if (t < E) v = L.mx[t];
        else {
            const int u = t - E;
            const double* r = sq + 5 * (U * U + u);
            v = (maxact ? maxact[u] : 1.0) * r[0] * r[2];
        }
        L.jm[t] = v;
        if (L.xm) L.xm[t] = v;
        if (writer) {
            g.wk.in_m[t] = v;
            if (g.tape) g.tape[(long)g.step * (D + D * D + E * D + E + E * E + D * E) + t] = v;
        }
    }
    for (int e = t; e < D * D; e += blockDim.x) {
        int c;
        const int r = idiv_s(e, D, c);
        double v;
        if (r < E && c < E) v = L.sx[r * E + c];
        else if (r < E) v = L.t1[(c - E) * E + r] * cdiag_of(c - E);
        else if (c < E) v = L.t1[(r - E) * E + c] * cdiag_of(r - E);
        else {
            const int u = r - E, v2 = c - E;
            const double* q = sq + 5 * (u * U + v2);
            const double val = (q[1] - q[0]) * q[3] - (q[2] - q[0]) * q[4];
            v = (maxact ? maxact[u] * maxact[v2] : 1.0) * val / 2.0;
        }
        L.js[e] = v;
        if (L.xs) L.xs[e] = v;
        if (writer) {
            g.wk.in_s[e] = v;
            if (r < E) s1_dst[r * D + c] = v;
            if (g.tape) {
                double* rec = g.tape + (long)g.step * (D + D * D + E * D + E + E * E + D * E);
                rec[D + e] = v;
                if (r < E) rec[D + D * D + r * D + c] = v;
            }
        }
    }
    __syncthreads();
}

// Reduce the tile / stream-K partials of the local pairs and the row-chunk partials of the
// owned outputs into this rank's segment (LDS copy + global gather buffer).  Four lanes per
// pair sum fixed quarters of the partial list and are combined in a fixed tree.
// Round `base` of mm_pack (4 threads per pair), split in two so that the loads of the first round are ISSUED at the very
// start of the glue kernel, together with its other loads, and consumed after them (vmcnt is in order: one round trip).
// Only kernel arguments go into the addresses (closed-form wave ranges).  Threads 0..255 do the pack whatever the
// workgroup size, so the summation order -- and with it every bit of the result -- is the same in both hosts.
struct PackPre {
    double v[16];
    double isdet;
};
// stream-K partials are slot-major: sk_part[slot][pair] (row stride sk_pls), slot = wave - first wave of the pair
// (unused slots stay zero).  Lane gq of a pair's four lanes sums the slots [gq * sk_maxw / 4, (gq + 1) * sk_maxw / 4) in
// order; for one slot the lanes of a wave read consecutive pairs, i.e. every load instruction touches a few whole
// cache lines (the pair-major layout of the first version cost 64 lines per instruction), and the addresses still come
// from the thread index alone.
__device__ __forceinline__ void mm_pack_issue(const MMWork& wk, int base, PackPre& pp) {
    const int t = threadIdx.x;
    const int k = base + (t >> 2), gq = t & 3;
    pp.isdet = 0.0;
#pragma unroll
    for (int u = 0; u < 16; ++u) pp.v[u] = 0.0;
    if (t >= 256 || k >= wk.PL) return;
    pp.isdet = wk.pair_isdet[k];
    if (wk.sk_waves > 0) {
        const int qw = wk.sk_maxw >> 2;
        const double* src = wk.sk_part + (long)(gq * qw) * wk.sk_pls + k;
#pragma unroll
        for (int u = 0; u < 16; ++u)
            if (u < qw) pp.v[u] = src[(long)u * wk.sk_pls];
    } else {   // tile partials (one-launch small step, reverse sweep, tiled pair kernel): the lane's first eight (value, trace) couples
        const double* part = wk.pair_part + (long)k * wk.NT * 2;
        const int q0 = (int)((long)wk.NT * gq / 4), q1 = (int)((long)wk.NT * (gq + 1) / 4);
#pragma unroll
        for (int u = 0; u < 8; ++u)
            if (q0 + u < q1) {
                pp.v[2 * u] = part[2 * (q0 + u)];
                pp.v[2 * u + 1] = part[2 * (q0 + u) + 1];
            }
    }
}
__device__ __forceinline__ void mm_pack_sum(const MMWork& wk, int base, const PackPre& pp, double& s0, double& s1) {
    const int t = threadIdx.x;
    const int k = base + (t >> 2), gq = t & 3;
    s0 = 0.0;
    s1 = 0.0;
    if (t >= 256 || k >= wk.PL) return;
    if (wk.sk_waves > 0) {
#pragma unroll
        for (int u = 0; u < 16; ++u) s0 += pp.v[u];
        const int qw = wk.sk_maxw >> 2;
        const double* src = wk.sk_part + (long)(gq * qw) * wk.sk_pls + k;
        for (int u0 = 16; u0 < qw; u0 += 8) {   // few pairs spread over many waves (a rank of a sharded model): eight requests
            double w8[8];                        // together, the sums in slot order (one at a time: 33 us of pack at W = 8)
#pragma unroll
            for (int u = 0; u < 8; ++u) w8[u] = src[(long)min(u0 + u, qw - 1) * wk.sk_pls];
#pragma unroll
            for (int u = 0; u < 8; ++u) s0 += (u0 + u < qw) ? w8[u] : 0.0;
        }
    } else {
        const double* part = wk.pair_part + (long)k * wk.NT * 2;
        const int q0 = (int)((long)wk.NT * gq / 4), q1 = (int)((long)wk.NT * (gq + 1) / 4);
#pragma unroll
        for (int u = 0; u < 8; ++u)   // (requested by mm_pack_issue; same order of additions as the plain loop)
            if (q0 + u < q1) {
                s0 += pp.v[2 * u];
                s1 += pp.v[2 * u + 1];
            }
        for (int q = q0 + 8; q < q1; ++q) {
            s0 += part[2 * q];
            s1 += part[2 * q + 1];
        }
    }
}

__device__ __forceinline__ void mm_pack(const MMWork& wk, int D, int E, const GlueLds& L, PackPre& pp, bool writer) {
    const int t = threadIdx.x;
    double* seg = wk.gath + (long)wk.rank * wk.SEG;
    for (int base = 0; base < wk.PL; base += 64) {
        const int k = base + (t >> 2), gq = t & 3;
        if (base > 0) mm_pack_issue(wk, base, pp);
        double s0, s1;
        mm_pack_sum(wk, base, pp, s0, s1);
        s0 += __shfl_xor(s0, 1);
        s1 += __shfl_xor(s1, 1);
        s0 += __shfl_xor(s0, 2);
        s1 += __shfl_xor(s1, 2);
        if (t < 256 && k < wk.PL && gq == 0) {
            const bool diag = k * wk.nranks + wk.rank < E;     // diagonal pairs come first in the dealing order
            const double v = (diag ? (s0 - s1) : s0) * pp.isdet;   // mgpr.py:144-145
            if (writer) seg[k] = v;
            L.seg[k] = v;
        }
    }
    const int W1 = 1 + D;
    for (int e = t; e < wk.EL * W1; e += blockDim.x) {   // M_a and V_a: sums of the chunk contributions
        int idx;
        const int o = idiv_s(e, W1, idx);
        double sum = 0.0;
        _Pragma("unroll 8") for (int ch = 0; ch < wk.NCHM; ++ch) sum += L.mp[(o * wk.NCHM + ch) * W1 + idx];
        if (writer) seg[wk.OUTOFF + e] = sum;
        L.seg[wk.OUTOFF + e] = sum;
    }
    __syncthreads();
}

// packed results -> out_M [E], out_S [E][E], out_V [D][E] (writer); always left in LDS (oM, oS, oV)
__device__ __forceinline__ void mm_assemble(const MMWork& wk, const double* src, const double* var, int D, int E, double* oM,
                            double* oS, double* oV, bool writer) {
    const int t = threadIdx.x;
    for (int a = t; a < E; a += blockDim.x) {
        const double v = src[(a % wk.nranks) * wk.SEG + wk.OUTOFF + (a / wk.nranks) * (1 + D)];
        oM[a] = v;
        if (writer) wk.out_M[a] = v;
    }
    for (int e = t; e < D * E; e += blockDim.x) {
        int a;
        const int d = idiv_s(e, E, a);
        const double v = src[(a % wk.nranks) * wk.SEG + wk.OUTOFF + (a / wk.nranks) * (1 + D) + 1 + d];
        oV[e] = v;
        if (writer) wk.out_V[e] = v;
    }
    __syncthreads();
    for (int e = t; e < E * E; e += blockDim.x) {
        int b;
        const int a = idiv_s(e, E, b);
        const int hi = a > b ? a : b, lo = a > b ? b : a;
        const int kk = pair_order_index(E, hi, lo);
        double v = src[(kk % wk.nranks) * wk.SEG + kk / wk.nranks];
        if (a == b) v += var[a];                                   // mgpr.py:146
        v = fma(-oM[a], oM[b], v);                                 // mgpr.py:147
        oS[e] = v;
        if (writer) wk.out_S[e] = v;
    }
    __syncthreads();
}

// ------------------------------------------------------------------ peer exchange (see GlueArgs::xq, moment.h)
// System-scope accesses throughout: the words are written by other GPUs (or other processes' kernels on this GPU) while
// this kernel runs, so nothing here may be served from a non-coherent cache.
__device__ __forceinline__ unsigned long long xq_ld(const unsigned long long* p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
// wait until every rank's flag in this exchange's slot has reached the epoch (thread r watches rank r); a wait that
// runs out of patience records the epoch in the area's error word and goes on -- the host reports it, nothing hangs
__device__ __forceinline__ void xq_wait(const GlueArgs& g) {
    const int t = threadIdx.x, W = g.xq_W;
    if (t < W) {
        const unsigned long long epoch = xq_ld(g.xq) + (unsigned long long)g.xq_k + 1ULL;
        const unsigned long long* flag = g.xq + 8 + (int)(epoch & 1ULL) * W + t;
        int it = 0;
        while (__hip_atomic_load(flag, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) < epoch) {
            if (++it > g.xq_spin) {
                __hip_atomic_store(g.xq + 1, epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                break;
            }
            __builtin_amdgcn_s_sleep(8);
        }
    }
    __syncthreads();
}
// the W segments of this exchange, [rank][SEG] as the gather buffer holds them, into LDS
__device__ __forceinline__ void xq_load_segments(const GlueArgs& g, double* seg_lds) {
    const int W = g.xq_W, SEG = g.wk.SEG;
    const unsigned long long epoch = xq_ld(g.xq) + (unsigned long long)g.xq_k + 1ULL;
    const unsigned long long* data = g.xq + xq_data_off(W) + (size_t)(epoch & 1ULL) * W * g.xq_cap;
    for (int e = threadIdx.x; e < W * SEG; e += blockDim.x) {
        int i;
        const int r = idiv_s(e, SEG, i);
        seg_lds[e] = __longlong_as_double((long long)xq_ld(data + (size_t)r * g.xq_cap + i));
    }
}
// deliver this rank's segment (in LDS after mm_pack) to every rank's area, then raise this rank's flag there
__device__ __forceinline__ void xq_push(const GlueArgs& g, const double* seg_lds) {
    const int t = threadIdx.x, W = g.xq_W, SEG = g.wk.SEG, me = g.wk.rank;
    const unsigned long long epoch = xq_ld(g.xq) + (unsigned long long)g.xq_k + 1ULL;
    const size_t off = (size_t)xq_data_off(W) + ((size_t)(epoch & 1ULL) * W + me) * g.xq_cap;
    for (int e = t; e < W * SEG; e += blockDim.x) {
        int i;
        const int r = idiv_s(e, SEG, i);
        __hip_atomic_store(g.xq_peers[r] + off + i, (unsigned long long)__double_as_longlong(seg_lds[i]), __ATOMIC_RELAXED,
                           __HIP_MEMORY_SCOPE_SYSTEM);
    }
    __threadfence_system();
    __syncthreads();
    if (t < W)
        __hip_atomic_store(g.xq_peers[t] + 8 + (int)(epoch & 1ULL) * W + me, epoch, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

// ------------------------------------------------------------------ RbfController inside the link
// controllers.py:108-121 -- M, S, V = predict_given_factorizations(m, s, 0 * iK, beta) of the policy GP (mgpr.py:99-149)
// -- evaluated by the workgroup that runs the link, from the state in LDS, instead of by an operand launch and a pair
// launch of the policy GP's own: an RbfController has 10-50 basis functions (examples/: bf = 10, 30, 40), its O(bf^2)
// sums are microseconds of one workgroup's time, and the two launches it used to cost were 40 % of a step at the sizes
// PILCO is used at (profiles/r03: policy head 11.8 us + policy pair kernel 4.8 us of a 41.7 us step).
// Same formulas as prep_device.h / pair_device.h (their comments cite the reference lines), plain fp64 `exp`.
// Every reduction is done by threads 0..255 in a fixed order, whatever the workgroup size: all workgroups of a head -- and
// k_glue -- produce the same bits.
struct RbfInlineLayout {
    int ctr, bet, il, var, lvar, aug0, aug1, piv, T, Q, det, pt, red, red3, total;
};
__host__ __device__ inline RbfInlineLayout rbf_inline_layout(int E, int U, int bf) {
    const int P = U * (U + 1) / 2, nmat = U + P;
    RbfInlineLayout l;
    int o = 0;
    l.ctr = o; o += bf * E;                 // centred centres  zeta_i = c_i - m
    l.bet = o; o += U * bf;                 // beta of every output
    l.il = o;  o += U * E;                  // 1 / lengthscale
    l.var = o; o += U;
    l.lvar = o; o += U;                     // log of the signal variances (phase 3's exponents)
    l.aug0 = o; o += nmat * E * 2 * E;      // augmented matrices of the batched Gauss-Jordan (ping)
    l.aug1 = o; o += nmat * E * 2 * E;      //                                                (pong)
    l.piv = o; o += nmat * E;               // pivots -> determinants
    l.T = o;   o += U * E * E;              // T_u = (s + Lambda_u^2)^-1
    l.Q = o;   o += P * E * E;              // Q_uv = R_uv^-1 s / 2
    l.det = o; o += nmat;                   // det B_u | det R_uv
    l.pt = o;  o += bf * (2 * E + 2) + bf * 16;   // per point of the current pair: u_i, v_i, p_i = 2 Q z_i, w_i | 16 doubles of scratch per point
    l.red = o; o += 4 * (E + 2);            // wave partials (mean part)
    l.red3 = o; o += 4;                     // wave partials (covariance part: it may run beside the mean part)
    l.total = (o + 1) & ~1;
    return l;
}

// The policy GP's constants (raw centres, transposed as stored: [E][bf]; beta [U][bf]; lengthscales [U][E]; variances [U])
// into LDS.  Issued together with the link's first batch of loads, so that their memory round trip is the link's own.
// Two halves: the loads of the first round (8 elements per thread: every controller of the reference's examples is one round)
// are REQUESTED before the link's own batch and stored behind it -- one memory round trip for both.
struct RbfPre {
    double v[8];
};
__device__ __forceinline__ void rbf_policy_preload_round(const GlueArgs& g, const GlueLds& L, int base, RbfPre& pre, bool issue, bool commit) {
    const int E = g.E, U = g.U, t = threadIdx.x, nthr = blockDim.x;
    const MMModel& pm = g.pmd;
    const int bf = pm.n, np = pm.npad;
    const RbfInlineLayout lay = rbf_inline_layout(E, U, bf);
    double* W = L.pol;
    const int n0 = bf * E, n1 = n0 + U * bf, n2 = n1 + U * E, total = n2 + U;
    if (issue) {
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int e = base + k * nthr + t;
            pre.v[k] = 0.0;
            if (base + k * nthr >= total) continue;   // (workgroup-uniform: a 10-basis-function controller is one element per thread)
            if (e < n1) {
                int i;
                const int row = idiv_s(e < n0 ? e : e - n0, bf, i);
                pre.v[k] = (e < n0 ? pm.Pt : pm.beta)[(long)row * np + i];
            } else if (e < n2) pre.v[k] = pm.ls[e - n1];
            else if (e < total) pre.v[k] = pm.var[e - n2];
        }
    }
    if (commit) {
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int e = base + k * nthr + t;
            if (base + k * nthr >= total) continue;
            if (e < n0) W[lay.ctr + e] = pre.v[k];
            else if (e < n1) W[lay.bet + (e - n0)] = pre.v[k];
            else if (e < n2) W[lay.il + (e - n1)] = pre.v[k];
            else if (e < total) W[lay.var + (e - n2)] = pre.v[k];
        }
    }
}
__device__ __forceinline__ void rbf_policy_preload_rest(const GlueArgs& g, const GlueLds& L, RbfPre& pre) {
    const int total = g.pmd.n * g.E + g.U * g.pmd.n + g.U * g.E + g.U, nthr = blockDim.x;
    rbf_policy_preload_round(g, L, 0, pre, false, true);
    for (int base = 8 * nthr; base < total; base += 8 * nthr) rbf_policy_preload_round(g, L, base, pre, true, true);
}

__device__ __forceinline__ void rbf_policy_inline(const GlueArgs& g, const GlueLds& L) {
    const int E = g.E, U = g.U, t = threadIdx.x, nthr = blockDim.x;
    const MMModel& pm = g.pmd;
    const int bf = pm.n, np = pm.npad;
    const int P = U * (U + 1) / 2, nmat = U + P;
    const RbfInlineLayout lay = rbf_inline_layout(E, U, bf);
    double* W = L.pol;
    double *ctr = W + lay.ctr, *bet = W + lay.bet, *il = W + lay.il, *var = W + lay.var, *piv = W + lay.piv, *Tm = W + lay.T,
           *Qm = W + lay.Q, *det = W + lay.det, *pt = W + lay.pt, *red = W + lay.red;
    const double* mx = L.mx;
    const double* sx = L.sx;   // [E][E]
    // ---- 0. the model is in LDS already (rbf_policy_preload, with the link's first loads): centre the points on the
    //         state this link has just produced, lengthscales -> reciprocals.  ctr is [E][bf] (as the centres are stored).
    (void)np;
    for (int e = t; e < bf * E; e += nthr) {
        int i;
        ctr[e] -= mx[idiv_s(e, bf, i)];
    }
    for (int e = t; e < U * E; e += nthr) il[e] = 1.0 / il[e];
    __syncthreads();
    // ---- 1. all D x D systems at once: [B_u | I] (mean part, mgpr.py:103-111) and [R_uv | s] (mgpr.py:121-129),
    //         unpivoted Gauss-Jordan (B is SPD, R diagonally similar to an SPD matrix), one barrier per pivot for all
    const int nc = 2 * E, msz = E * nc;
    double* cur = W + lay.aug0;
    double* nxt = W + lay.aug1;
    for (int e = t; e < nmat * msz; e += nthr) {
        int rc, c;
        const int q = idiv_s(e, msz, rc), r = idiv_s(rc, nc, c);
        double v;
        if (q < U) {
            if (c < E) v = fma(sx[r * E + c], il[q * E + r] * il[q * E + c], (r == c) ? 1.0 : 0.0);
            else v = (c - E == r) ? 1.0 : 0.0;
        } else {
            int a = 0, pq = q - U;
            while ((a + 1) * (a + 2) / 2 <= pq) ++a;          // pair index pq = a (a + 1) / 2 + b, a >= b
            const int b = pq - a * (a + 1) / 2;
            if (c < E) {
                const double ia = il[a * E + c], ib = il[b * E + c];
                v = fma(sx[r * E + c], ia * ia + ib * ib, (r == c) ? 1.0 : 0.0);
            } else {
                v = sx[r * E + (c - E)];
            }
        }
        cur[e] = v;
    }
    int q0_, rc0_, r0_, c0_;   // the thread's first element (all of them for U + P <= nthr / (2 E^2) systems): indices made once, not per pivot
    q0_ = idiv_s(t, msz, rc0_);
    r0_ = idiv_s(rc0_, nc, c0_);
    for (int k = 0; k < E; ++k) {
        __syncthreads();
        for (int e = t; e < nmat * msz; e += nthr) {
            int q = q0_, rc = rc0_, r = r0_, c = c0_;
            if (e != t) {
                q = idiv_s(e, msz, rc);
                r = idiv_s(rc, nc, c);
            }
            const double* Mq = cur + q * msz;
            const double pk = Mq[k * nc + c] * fast_rcp(Mq[k * nc + k]);   // (as the register Gauss-Jordan of the operand kernel)
            nxt[e] = (r == k) ? pk : fma(-Mq[r * nc + k], pk, Mq[rc]);
            if (rc == 0) piv[q * E + k] = Mq[k * nc + k];
        }
        double* tmp = cur;
        cur = nxt;
        nxt = tmp;
    }
    __syncthreads();
    if (t < nmat) {
        double d = 1.0;
        for (int k = 0; k < E; ++k) d *= piv[t * E + k];
        det[t] = d;
    }
    for (int e = t; e < U * E * E; e += nthr) {   // T_u = Lambda^-1 B^-1 Lambda^-1
        int rc, c;
        const int u = idiv_s(e, E * E, rc), r = idiv_s(rc, E, c);
        Tm[e] = cur[u * msz + r * nc + E + c] * il[u * E + r] * il[u * E + c];
    }
    // (the logs the covariance part needs, by the workgroup's last threads: they run beside the T / Q products)
    if (t >= nthr - U) W[lay.lvar + (t - (nthr - U))] = log(var[t - (nthr - U)]);
    for (int e = t; e < P * E * E; e += nthr) {   // Q_uv = R^-1 s / 2
        int rc, c;
        const int pq = idiv_s(e, E * E, rc), r = idiv_s(rc, E, c);
        Qm[e] = 0.5 * cur[(U + pq) * msz + r * nc + E + c];
    }
    __syncthreads();
    const int lane = t & 63, w = t >> 6;
    // ---- 2. mean and input-output covariance of every output (mgpr.py:113-118) and 3. covariance of every output pair
    //         (mgpr.py:120-147 with iK = 0).  bf <= 256: thread i < bf owns point i; plain run-time loops over LDS (no unrolled
    //         register arrays: this code sits in the serial link's instruction stream, where its SIZE costs as much as its work).
    //         Both parts need the batched Gauss-Jordan only, and pair k needs the means of outputs <= k: in a 512-thread host
    //         (the fused head) round k evaluates output k on the lower half of the workgroup and pair k BESIDE it on the upper
    //         half -- three barrier intervals per round instead of six, the pair's row operations hidden behind the mean
    //         part's exp.  A 256-thread host (k_glue) runs the same code one part after the other: same arithmetic per
    //         element, same bits.
    const bool par = nthr >= 512 && 2 * bf <= 256;
    double* red3 = W + lay.red3;
    const int rounds = par ? (U > P ? U : P) : U + P;
    for (int k = 0; k < rounds; ++k) {
        const int u = k;                              // output of this round's mean part
        const bool do2 = k < U;
        const int pq = par ? k : k - U;               // pair of this round's covariance part
        const bool do3 = pq >= 0 && pq < P;
        int a = 0, b = 0;
        if (do3) {
            while ((a + 1) * (a + 2) / 2 <= pq) ++a;  // pair index pq = a (a + 1) / 2 + b, a >= b
            b = pq - a * (a + 1) / 2;
        }
        const double* Q = Qm + (do3 ? pq : 0) * E * E;
        double* uv = pt;                 // [bf] u_i (row side, output a)
        double* vv = pt + bf;            // [bf] v_j (column side, output b)
        double* pv = pt + 2 * bf;        // [bf][E] 2 Q z_i
        double* wv = pv + bf * E;        // [bf][E] w_j
        // -- interval A: the points' exponentials and their wave sums | the pair's row and column operands
        if (do2) {
            double lb = 0.0;
            if (t < bf) {
                double q = 0.0;
                for (int r = 0; r < E; ++r) {
                    double tz = 0.0;
                    for (int c = 0; c < E; ++c) tz = fma(Tm[(u * E + r) * E + c], ctr[c * bf + t], tz);
                    q = fma(ctr[r * bf + t], tz, q);
                }
                lb = exp(-0.5 * q) * bet[u * bf + t];
            }
            if (t < 256) {
                const double gs = wave_sum_lane63(lb);
                if (lane == 63) red[w * (E + 2)] = gs;
                for (int d = 0; d < E; ++d) {
                    const double v = wave_sum_lane63((t < bf) ? ctr[d * bf + t] * lb : 0.0);
                    if (lane == 63) red[w * (E + 2) + 1 + d] = v;
                }
            }
        }
        if (do3) {
            const double la = W[lay.lvar + a], lb_ = W[lay.lvar + b];
            const int t3 = par ? t - 256 : t;
            for (int i = t3; i >= 0 && i < 2 * bf; i += nthr) {
                const int side = i >= bf, ii = side ? i - bf : i;
                const double* ilo = il + (side ? b : a) * E;
                double* xrow = wv + ii * E;          // side 1: w_j is what stays here; side 0: scratch in the OTHER half below
                if (!side) xrow = pv + ii * E;       // (z_i parks in the row that 2 Q z_i will overwrite: read completely first)
                double kk = side ? lb_ : la;
                for (int d = 0; d < E; ++d) {
                    const double zd = ctr[d * bf + ii];
                    const double xd = zd * ilo[d] * ilo[d];
                    xrow[d] = xd;
                    kk = fma(-0.5 * zd, xd, kk);
                }
                double quad = 0.0;
                if (side) {
                    for (int r = 0; r < E; ++r) {
                        double y = 0.0;
                        for (int c = 0; c < E; ++c) y = fma(Q[r * E + c], xrow[c], y);
                        quad = fma(xrow[r], y, quad);
                    }
                    vv[ii] = kk + quad;
                } else {
                    // y = Q z needs all of z while 2 y overwrites it: through the point's 16 doubles of scratch behind the vectors
                    double* ytmp = pt + bf * (2 * E + 2) + ii * 16;
                    for (int r = 0; r < E; ++r) {
                        double y = 0.0;
                        for (int c = 0; c < E; ++c) y = fma(Q[r * E + c], xrow[c], y);
                        quad = fma(xrow[r], y, quad);
                        ytmp[r] = 2.0 * y;
                    }
                    for (int r = 0; r < E; ++r) xrow[r] = ytmp[r];
                    uv[ii] = kk + quad;
                }
            }
        }
        __syncthreads();
        // -- interval B: the mean part's totals | the pair's bf^2 exponentials and their wave sums
        if (do2 && t <= E) red[t] = ((red[t] + red[(E + 2) + t]) + red[2 * (E + 2) + t]) + red[3 * (E + 2) + t];   // (row 0 receives the totals)
        if (do3 && t < 256) {
            double acc = 0.0;
            for (int idx = t; idx < bf * bf; idx += 256) {
                int j;
                const int i = idiv_s(idx, bf, j);
                double e = uv[i] + vv[j];
                for (int d = 0; d < E; ++d) e = fma(pv[i * E + d], wv[j * E + d], e);
                acc = fma(bet[a * bf + i] * bet[b * bf + j], exp(e), acc);
            }
            const double sacc = wave_sum_lane63(acc);
            if (lane == 63) red3[w] = sacc;
        }
        __syncthreads();
        // -- interval C: M_u and V_u | S_ab (thread 0 has written the means of outputs <= k by then: a <= pq = k)
        if (do2) {
            const double cu = var[u] / sqrt(det[u]);
            if (t == 0) L.mu[u] = cu * red[0];
            if (t < E) {
                double acc = 0.0;
                for (int kq = 0; kq < E; ++kq) acc = fma(Tm[(u * E + t) * E + kq], red[1 + kq], acc);
                L.cxu[t * U + u] = cu * acc;   // V (E, U)
            }
        }
        if (do3 && t == 0) {
            const double N = (red3[0] + red3[1]) + (red3[2] + red3[3]);
            double v = N / sqrt(det[U + pq]);
            if (a == b) v += var[a];                         // mgpr.py:146
            v = fma(-L.mu[a], L.mu[b], v);                   // mgpr.py:147
            L.su[a * U + b] = v;
            L.su[b * U + a] = v;
        }
        __syncthreads();
    }
}

// The serial link.  On return (GF_POLICY) the joint Gaussian is in L.jm / L.js and the (propagated) state in L.mx / L.sx.
// All threads of the workgroup must call it; `writer` selects the one workgroup that stores results to global memory.
// PK selects which controller code is COMPILED INTO the host kernel: -1 all of it (k_glue: dispatch at run time), 0 no
// controller (control_dim 0), 3 LinearController, 1 RbfController reduced from its own launches (GF_RBF_POST), 2
// RbfController evaluated inline; SR (single rank): the peer-exchange waits / stores and the general (gathered-segments)
// assemble + propagate are left out as well -- one rank always takes the two-phase `fast` path below.  The
// link is a chain of latencies through straight-line code: round 3 measured +1.5 us (no policy) to +3 us (linear policy)
// per step on the fused head when the inline-RBF code was merely PRESENT in its instruction stream (instruction fetch), so
// every fused head is instantiated for the controller kind it serves.
template <int PK = -1, bool SR = false>
__device__ __forceinline__ void glue_body(const GlueArgs& g, const GlueLds& L, bool writer) {
    const int E = g.E, D = g.D, U = g.U, t = threadIdx.x;
    int mp_n = (g.flags & GF_PACK) ? g.wk.EL * g.wk.NCHM * (1 + D) : 0;
    int seg_n = (g.flags & (GF_PACK | GF_ASSEMBLE)) ? g.wk.SEG * ((g.flags & GF_PACK) ? 1 : g.wk.nranks) : 0;
    if (g.flags & GF_RBF_POST) {  // this launch reduces the POLICY GP (inputs = state, outputs = controls)
        mp_n = g.pwk.EL * g.pwk.NCHM * (1 + E);
        seg_n = g.pwk.SEG;
    }
    const bool dbg0 = (t == 0) && writer;
    const int dbo = g.dbg_off ? g.dbg_off : ((g.step == 0) ? 16 : 0);  // the initial glue of a rollout stamps slots 24..29
    DBG_STAMP(g.wk, 8 + dbo, dbg0);
    PackPre pp;   // first round of the pack: its loads are in flight together with the batch below
    if ((g.flags & GF_PACK) && !MM_ABL(g.wk, 16)) mm_pack_issue(g.wk, 0, pp);
    const bool xq_in = !SR && g.xq && (g.flags & GF_ASSEMBLE) && !(g.flags & GF_PACK);
    if constexpr (!SR)
        if (xq_in) xq_wait(g);   // (the segment loads below must not be issued before the flags have been seen)
    {   // one batch of loads for everything the serial part reads
        const bool need_state = (g.flags & (GF_PROPAGATE | GF_TRAJ | GF_POLICY | GF_RBF_PRE)) != 0;
        const bool lin = (g.flags & GF_POLICY) && g.pol_kind == PILCO_POLICY_LINEAR;
        const bool pol = (g.flags & GF_POLICY) && g.pol_kind != PILCO_POLICY_NONE;
        const LoadSeg sg[8] = {
            {0, g.m_x, need_state ? E : 0},
            {L.o_sx, g.s_x, need_state ? E * E : 0},
            {L.o_s1, g.s1, (g.flags & GF_PROPAGATE) ? E * D : 0},
            {L.o_mp, (g.flags & GF_RBF_POST) ? g.pwk.mean_part : g.wk.mean_part, mp_n},
            {L.o_seg, g.wk.gath, ((g.flags & GF_ASSEMBLE) && !(g.flags & GF_PACK) && !xq_in) ? seg_n : 0},
            {L.o_js, g.W, lin ? U * E : 0},   // W parks in the joint-covariance buffer until write_joint overwrites it
            {L.o_misc + 128, g.b, lin ? U : 0},                       // the policy's small vectors ride in the same batch: read
            {L.o_misc + 160, g.maxact, (pol && g.maxact) ? U : 0},    // from global memory later each costs a DRAM round trip
        };
        RbfPre rpre;
        const bool rbf_in = (PK < 0 || PK == 2) && (g.flags & GF_POLICY) && g.pol_kind == PILCO_POLICY_RBF && g.pol_inline;
        if constexpr (PK < 0 || PK == 2)
            if (rbf_in) rbf_policy_preload_round(g, L, 0, rpre, true, false);
        multi_load<8, 4>(L.mx, sg);
        if constexpr (PK < 0 || PK == 2)
            if (rbf_in) rbf_policy_preload_rest(g, L, rpre);
        if constexpr (!SR)
            if (xq_in) xq_load_segments(g, L.seg);
    }
    __syncthreads();

    DBG_STAMP(g.wk, 9 + dbo, dbg0);
    if ((g.flags & GF_PACK) && !MM_ABL(g.wk, 16)) mm_pack(g.wk, D, E, L, pp, writer);
    if constexpr (!SR)
        if ((g.flags & GF_PACK) && g.xq_peers && writer) xq_push(g, L.seg);
    DBG_STAMP(g.wk, 10 + dbo, dbg0);
    const bool fast = (g.flags & (GF_PACK | GF_ASSEMBLE | GF_PROPAGATE)) == (GF_PACK | GF_ASSEMBLE | GF_PROPAGATE) && g.wk.nranks == 1;
    // A head without a controller (PK = 0: control_dim 0, D = E): the joint Gaussian IS the propagated state, so the phase that
    // produces the state writes the joint too -- the copies write_joint would make one barrier interval later (same values).
    const bool joint_is_state = (PK == 0) && fast && (g.flags & GF_POLICY) && !g.act_out && U == 0;
    if (fast) {
        // The rollout's common case in two phases instead of five (every phase boundary is a workgroup barrier plus an LDS
        // round trip on this serial path).  Same operations in the same order as mm_assemble + GF_PROPAGATE below: the
        // results are bitwise identical.  S_rc (mgpr.py:143-147) and t1 = s1 V (pilco.py:149) straight from the segment:
        const double* seg = L.seg;
        const int W1 = 1 + D, OO = g.wk.OUTOFF;
        for (int e = t; e < E * E; e += blockDim.x) {
            int c;
            const int r = idiv_s(e, E, c);
            const int hi = r > c ? r : c, lo = r > c ? c : r;
            double v = seg[pair_order_index(E, hi, lo)];
            if (r == c) v += g.var[r];
            v = fma(-seg[OO + r * W1], seg[OO + c * W1], v);
            L.su[e] = v;
            double acc = 0.0;
            _Pragma("unroll 8") for (int k = 0; k < D; ++k) acc = fma(L.s1[r * D + k], seg[OO + c * W1 + 1 + k], acc);
            L.t1[e] = acc;
        }
        if (t < E) L.mu[t] = seg[OO + t * W1];
        if (writer && g.tape) {   // the tape wants the GP outputs (M, S, V) of the step: V in [D][E] order
            for (int e = t; e < D * E; e += blockDim.x) {
                int a_;
                const int d_ = idiv_s(e, E, a_);
                L.cxu[e] = seg[OO + a_ * W1 + 1 + d_];
            }
        }
        __syncthreads();
        if (writer && g.tape && g.step >= 1) {
            double* rec = g.tape + (long)(g.step - 1) * (D + D * D + E * D + E + E * E + D * E) + D + D * D + E * D;
            if (t < E) rec[t] = L.mu[t];
            for (int e = t; e < E * E; e += blockDim.x) rec[E + e] = L.su[e];
            for (int e = t; e < D * E; e += blockDim.x) rec[E + E * E + e] = L.cxu[e];
        }
        double* s1_dst = g.s1_out ? g.s1_out : g.s1;
        double* trec = g.tape ? g.tape + (long)g.step * (D + D * D + E * D + E + E * E + D * E) : nullptr;
        for (int e = t; e < E * E; e += blockDim.x) {   // in place: a thread reads sx only at the element it writes
            int c;
            const int r = idiv_s(e, E, c);
            const double v = ((L.su[e] + L.sx[e]) + L.t1[e]) + L.t1[c * E + r];
            L.sx[e] = v;
            if (writer) g.s_out[e] = v;
            if (joint_is_state) {   // (write_joint's stores for D = E, U = 0)
                L.js[e] = v;
                if (L.xs) L.xs[e] = v;
                if (writer) {
                    g.wk.in_s[e] = v;
                    s1_dst[e] = v;
                    if (trec) {
                        trec[D + e] = v;
                        trec[D + D * D + e] = v;
                    }
                }
            }
        }
        if (t < E) {
            const double v = L.mu[t] + L.mx[t];
            L.mx[t] = v;
            if (writer) g.m_out[t] = v;
            if (joint_is_state) {
                L.jm[t] = v;
                if (L.xm) L.xm[t] = v;
                if (writer) {
                    g.wk.in_m[t] = v;
                    if (trec) trec[t] = v;
                }
            }
        }
        __syncthreads();
    }
    if (!SR && !fast && (g.flags & GF_ASSEMBLE)) {
        // single rank: the LDS copy of the segment is the whole gather buffer
        mm_assemble(g.wk, L.seg, g.var, D, E, L.mu, L.su, L.cxu, writer);  // oM -> mu, oS -> su, oV -> cxu
        if (writer && g.tape && g.step >= 1) {
            double* rec = g.tape + (long)(g.step - 1) * (D + D * D + E * D + E + E * E + D * E) + D + D * D + E * D;
            if (t < E) rec[t] = L.mu[t];
            for (int e = t; e < E * E; e += blockDim.x) rec[E + e] = L.su[e];
            for (int e = t; e < D * E; e += blockDim.x) rec[E + E * E + e] = L.cxu[e];
        }
    }
    DBG_STAMP(g.wk, 11 + dbo, dbg0);
    if (!SR && !fast && (g.flags & GF_PROPAGATE)) {
        // t1 = s1 V (E,E); state += increment                      (pilco.py:147-149)
        for (int e = t; e < E * E; e += blockDim.x) {
            int c;
            const int r = idiv_s(e, E, c);
            double acc = 0.0;
            _Pragma("unroll 8") for (int k = 0; k < D; ++k) acc = fma(L.s1[r * D + k], L.cxu[k * E + c], acc);
            L.t1[e] = acc;
        }
        __syncthreads();
        for (int e = t; e < E * E; e += blockDim.x) {
            int c;
            const int r = idiv_s(e, E, c);
            const double v = ((L.su[e] + L.sx[e]) + L.t1[e]) + L.t1[c * E + r];
            L.t2[e] = v;
            if (writer) g.s_out[e] = v;
        }
        if (t < E) {
            const double v = L.mu[t] + L.mx[t];
            L.misc[96 + t] = v;
            if (writer) g.m_out[t] = v;
        }
        __syncthreads();
        for (int e = t; e < E * E; e += blockDim.x) L.sx[e] = L.t2[e];
        if (t < E) L.mx[t] = L.misc[96 + t];
        __syncthreads();
    }
    DBG_STAMP(g.wk, 12 + dbo, dbg0);
    if (writer && (g.flags & GF_TRAJ) && g.traj) {
        double* dst = g.traj + (long)g.step * (E + E * E);
        if (t < E) dst[t] = L.mx[t];
        for (int e = t; e < E * E; e += blockDim.x) dst[E + e] = L.sx[e];
    }
    if (writer && (g.flags & GF_RBF_PRE)) {  // RbfController: the state is the input of the policy GP (controllers.py:115-116)
        if (t < E) g.pwk.in_m[t] = L.mx[t];
        for (int e = t; e < E * E; e += blockDim.x) g.pwk.in_s[e] = L.sx[e];
    }
    if (g.flags & GF_POLICY) {
        const double* jcd = nullptr;   // the squash's diagonal, when write_joint is to apply it to c_xu
        bool lin_fused = false;        // linear controller + squash: the evaluations are in L.t2, write_joint_lin_squash combines them
        if (PK != 0 && PK != 3 && g.pol_kind == PILCO_POLICY_RBF) {
            // mean-function-only GP: iK = 0, then S -= diag(var - 1e-6)      (controllers.py:116-117)
            bool done = false;
            if constexpr (PK < 0 || PK == 2) {
                if (g.pol_inline) {
                    rbf_policy_inline(g, L);                                      // M (U), S (U,U), V (E,U) from the state in LDS
                    done = true;
                }
            }
            if constexpr (PK < 0 || PK == 1) {
                if (!done) {
                    PackPre pq;
                    mm_pack_issue(g.pwk, 0, pq);
                    mm_pack(g.pwk, E, U, L, pq, writer);
                    mm_assemble(g.pwk, L.seg, g.pvar, E, U, L.mu, L.su, L.cxu, writer);   // M (U), S (U,U), V (E,U)
                }
            }
            if (t < U) L.su[t * U + t] -= g.pvar[t] - 1e-6;
            __syncthreads();
            if (g.squash) {
                double* cdiag = L.misc + 1;
                squash_inplace(L, U, g.maxact ? L.misc + 160 : nullptr, cdiag);
                if (g.act_out) {
                    for (int e = t; e < E * U; e += blockDim.x) L.cxu[e] *= cdiag[e % U];
                    __syncthreads();
                } else {
                    jcd = cdiag;   // (write_joint scales on the fly)
                }
            }
        }
        if ((PK < 0 || PK == 3) && g.pol_kind == PILCO_POLICY_LINEAR) {
            // M = m W^T + b, S = W s W^T, V = W^T                  (controllers.py:52-54); W is in L.js (batched load)
            if (t < U) {
                double acc = L.misc[128 + t];
                _Pragma("unroll 8") for (int k = 0; k < E; ++k) acc = fma(L.js[t * E + k], L.mx[k], acc);
                L.mu[t] = acc;
            }
            for (int e = t; e < U * E; e += blockDim.x) {
                int c;
                const int u = idiv_s(e, E, c);
                double acc = 0.0;
                _Pragma("unroll 8") for (int k = 0; k < E; ++k) acc = fma(L.js[u * E + k], L.sx[k * E + c], acc);
                L.t1[e] = acc;  // W s
                L.cxu[c * U + u] = L.js[e];
            }
            __syncthreads();
            for (int e = t; e < U * U; e += blockDim.x) {
                int v;
                const int u = idiv_s(e, U, v);
                double acc = 0.0;
                _Pragma("unroll 8") for (int k = 0; k < E; ++k) acc = fma(L.t1[u * E + k], L.js[v * E + k], acc);
                L.su[e] = acc;
            }
            __syncthreads();
#ifdef LINK_STAMPS
            DBG_STAMP(g.wk, 36, dbg0);
#endif
            if (g.squash) {
                double* cdiag = L.misc + 1;  // [U]
                // the joint Gaussian is what follows and the evaluations fit t2: stop behind them (write_joint_lin_squash)
                lin_fused = !g.act_out && !joint_is_state && 5 * (U * U + U) <= L.nm * L.nm;
                squash_inplace(L, U, g.maxact ? L.misc + 160 : nullptr, cdiag, lin_fused ? L.t2 : nullptr, lin_fused);
                if (g.act_out) {
                    for (int e = t; e < E * U; e += blockDim.x) L.cxu[e] *= cdiag[e % U];   // V @ C, C diagonal
                    __syncthreads();
                } else {
                    jcd = cdiag;
                }
            }
#ifdef LINK_STAMPS
            DBG_STAMP(g.wk, 37, dbg0);
#endif
        }
        if (g.act_out) {
            if (writer) {
                if (t < U) g.act_out[t] = L.mu[t];
                for (int e = t; e < U * U; e += blockDim.x) g.act_out[U + e] = L.su[e];
                for (int e = t; e < E * U; e += blockDim.x) g.act_out[U + U * U + e] = L.cxu[e];
            }
        } else if (!joint_is_state) {
            if (lin_fused) write_joint_lin_squash(g, L, writer, L.t2, g.maxact ? L.misc + 160 : nullptr);
            else write_joint(g, L, writer, jcd);
        }
    }
    DBG_STAMP(g.wk, 13 + dbo, dbg0);
}

}  // namespace pilco
