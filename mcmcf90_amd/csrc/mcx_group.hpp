// mcx_group.hpp -- the lane-GROUP-per-chain step kernel (gfx950, wave64): MCMC_run / DRAM with the factors on chip.
//
// The lane-per-chain kernels of mcx_kernels.hpp stream every chain's packed factor R (and, with delayed rejection,
// R2 = R / drscale and iC TWICE) from HBM at every iteration although none of them changes between two adaptations
// (MCMC_adapt.F90:45-46).  Here one chain belongs to the 16 lanes of one DPP row -- four chains per wave -- and
//   * lane l of a group OWNS the columns l, l + 16, ... of R and R2 (rows 0..column: MCMC_propose's dtrmv('U','T'),
//     matutils.F90:108-109, one fma chain per column, rows ascending) and the rows l, l + 16, ... of the symmetric iC
//     (MCMC_DR_alpha13's dsymv, MCMC_DRAM.F90:180-182, one fma chain per row, columns ascending), all of it in REGISTERS
//     for the whole launch: an iteration reads no factor from memory at all;
//   * a vector element x_i that every column / row chain needs comes from lane i mod 16 by `row_newbcast` (the DPP
//     control the f64 pipe supports: v_fmac_f64_dpp = one instruction per term);
//   * the polar attempts of normal_bm (mcmcrand.F90:166-190) of one chain run SIXTEEN AT A TIME, attempt a of a round on
//     lane a: Philox is counter-based, so lane a computes the block that holds the attempt's two uniforms; a prefix count
//     over the group's accept ballot puts every accepted pair at the place the one-at-a-time loop would have given it,
//     and the attempts behind the one that completes the vector are dropped with their uniforms undrawn -- stream
//     position, deviates and the cached second deviate are those of the reference's loop;
//   * sums that the reference takes in index order across what are now lanes (priorfun, the banana target, the q-sums
//     of the quadratic forms) run as chains of `row_newbcast` adds / fmas in that order.
// Every chain of floating-point operations is the one of step_body / dr_body (same operands, same order), so the
// results are theirs bit for bit (tests/test_gpu_group.py compares the two kernel families and the oracle).
//
// Layout of a wave: lane = 16 * row + l16; chain slot = 4 * blockIdx.x + row (tile = slot / 64, lane-in-tile = slot % 64
// of the tile-interleaved global arrays).  D4 = npar rounded up to a multiple of four (template parameter: the register
// arrays are sized by it); NS = ceil(D4 / 16) slots per lane; positions npar..D4-1 are padding (zero everywhere).
#pragma once
#include "mcx_kernels.hpp"
#include <type_traits>

namespace mcx {

// Cross-lane exchange through LDS inside ONE wave (every block of this file is a single wave): the hardware retires a wave's LDS
// operations in order, so a read issued after a write sees it; what has to be said is the COMPILER's order between the write phase and the
// read phase -- an acquire-release fence at wavefront scope (no instruction) plus the wave barrier (no instruction either).  Without it
// the order rests on the alias analysis not proving the two index expressions distinct.
#define MCX_WAVE_LDS_SYNC() do { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier(); } while (0)

template <int I, int N, typename F>
MCX_DEV void sfor(F &&f)
{
    if constexpr (I < N) { f(std::integral_constant<int, I>{}); sfor<I + 1, N>(f); }
}

// ---------------------------------------------------------------- cross-lane primitives inside a row of 16 lanes
// value of lane N of the row, through the builtin (32-bit halves): hazards and scheduling are the compiler's
template <int N>
MCX_DEV double row_bcast(double x)
{
    int lo = __double2loint(x), hi = __double2hiint(x);
    lo = __builtin_amdgcn_update_dpp(0, lo, 0x150 + N, 0xf, 0xf, false);     // row_newbcast:N
    hi = __builtin_amdgcn_update_dpp(0, hi, 0x150 + N, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
}
// value of lane l16 + N of the same row (garbage past the row's end: callers mask)
template <int N>
MCX_DEV double row_down(double x) { return __shfl_down(x, N, 16); }
// GW lanes per chain: 16 (a DPP row) or 4 (a quad: sixteen chains per wave -- small npar with the chip full, where sixteen lanes would
// mostly idle).  Value of lane N of the group: row_newbcast, or quad_perm [N,N,N,N] on the two halves.
template <int GW, int N>
MCX_DEV double grp_bcast(double x)
{
    if constexpr (GW == 16) return row_bcast<N>(x);
    else {
        int lo = __double2loint(x), hi = __double2hiint(x);
        lo = __builtin_amdgcn_update_dpp(0, lo, N * 0x55, 0xf, 0xf, false);
        hi = __builtin_amdgcn_update_dpp(0, hi, N * 0x55, 0xf, 0xf, false);
        return __hiloint2double(hi, lo);
    }
}

// The hot sequences are single asm statements: v_fmac_f64_dpp / v_add_f64_dpp / v_mov_b64_dpp with row_newbcast (the one DPP
// control the f64 pipe takes).  The compiler's hazard recogniser does not look into inline asm, and a DPP operand must not be
// read within two wait states of the VALU instruction that wrote it: every statement opens with s_nop 1 (two wait states), and the
// instructions inside one statement never write a register that a later DPP operand of the same statement reads.  (The other DPP
// hazard -- five wait states after a VALU write of EXEC, i.e. v_cmpx -- does not arise: on gfx9 the compiler forms exec masks with
// v_cmp + s_and_saveexec, and tests/test_cabi_exports.py::test_group_kernels_hold_no_v_cmpx looks at the shipped code object.)
#define MCX_DPPC(i) " row_newbcast:" #i " row_mask:0xf bank_mask:0xf\n\t"
#define MCX_GFM(i, r) "v_fmac_f64_dpp %0, %1, %" #r MCX_DPPC(i)
#define MCX_GFM4(b) MCX_GFM(0, 2) MCX_GFM(1, 3) MCX_GFM(2, 4) MCX_GFM(3, 5)
#define MCX_GFM8(b) MCX_GFM4(b) MCX_GFM(4, 6) MCX_GFM(5, 7) MCX_GFM(6, 8) MCX_GFM(7, 9)
#define MCX_GFM12(b) MCX_GFM8(b) MCX_GFM(8, 10) MCX_GFM(9, 11) MCX_GFM(10, 12) MCX_GFM(11, 13)
#define MCX_GFM16(b) MCX_GFM12(b) MCX_GFM(12, 14) MCX_GFM(13, 15) MCX_GFM(14, 16) MCX_GFM(15, 17)
#define MCX_R4(r) "v"(r[0]), "v"(r[1]), "v"(r[2]), "v"(r[3])
#define MCX_R8(r) MCX_R4(r), "v"(r[4]), "v"(r[5]), "v"(r[6]), "v"(r[7])
#define MCX_R12(r) MCX_R8(r), "v"(r[8]), "v"(r[9]), "v"(r[10]), "v"(r[11])
#define MCX_R16(r) MCX_R12(r), "v"(r[12]), "v"(r[13]), "v"(r[14]), "v"(r[15])
// p = fma(r[i], x_i, p) for i = 0..N-1 in that order, x_i = lane i's z
template <int N>
MCX_DEV void blk_fmac(double &p, double z, const double *r)
{
    static_assert(N == 4 || N == 8 || N == 12 || N == 16, "blocks of four");
    if constexpr (N == 4) asm("s_nop 1\n\t" MCX_GFM4(0) : "+v"(p) : "v"(z), MCX_R4(r));
    else if constexpr (N == 8) asm("s_nop 1\n\t" MCX_GFM8(0) : "+v"(p) : "v"(z), MCX_R8(r));
    else if constexpr (N == 12) asm("s_nop 1\n\t" MCX_GFM12(0) : "+v"(p) : "v"(z), MCX_R12(r));
    else asm("s_nop 1\n\t" MCX_GFM16(0) : "+v"(p) : "v"(z), MCX_R16(r));
}
// two chains over the same coefficients: pa = fma(r[i], xa_i, pa), pb = fma(r[i], xb_i, pb).  (A chain whose first term is a plain
// product -- quadform_sym's `(i == 0) ? sij * dxi : dfma(...)` -- starts from -0.0: fma(a, b, -0.0) is a * b bit for bit, signed zeros
// included; of the f64 VALU operations only v_fmac_f64 and the VOP1 forms have a DPP encoding on gfx950.)
#define MCX_GF2(i, r) "v_fmac_f64_dpp %0, %2, %" #r MCX_DPPC(i) "v_fmac_f64_dpp %1, %3, %" #r MCX_DPPC(i)
#define MCX_GF2_4 MCX_GF2(0, 4) MCX_GF2(1, 5) MCX_GF2(2, 6) MCX_GF2(3, 7)
#define MCX_GF2_8 MCX_GF2_4 MCX_GF2(4, 8) MCX_GF2(5, 9) MCX_GF2(6, 10) MCX_GF2(7, 11)
#define MCX_GF2_12 MCX_GF2_8 MCX_GF2(8, 12) MCX_GF2(9, 13) MCX_GF2(10, 14) MCX_GF2(11, 15)
#define MCX_GF2_16 MCX_GF2_12 MCX_GF2(12, 16) MCX_GF2(13, 17) MCX_GF2(14, 18) MCX_GF2(15, 19)
template <int N>
MCX_DEV void blk_fmac2(double &pa, double &pb, double za, double zb, const double *r)
{
    static_assert(N == 4 || N == 8 || N == 12 || N == 16, "blocks of four");
    if constexpr (N == 4) asm("s_nop 1\n\t" MCX_GF2_4 : "+v"(pa), "+v"(pb) : "v"(za), "v"(zb), MCX_R4(r));
    else if constexpr (N == 8) asm("s_nop 1\n\t" MCX_GF2_8 : "+v"(pa), "+v"(pb) : "v"(za), "v"(zb), MCX_R8(r));
    else if constexpr (N == 12) asm("s_nop 1\n\t" MCX_GF2_12 : "+v"(pa), "+v"(pb) : "v"(za), "v"(zb), MCX_R12(r));
    else asm("s_nop 1\n\t" MCX_GF2_16 : "+v"(pa), "+v"(pb) : "v"(za), "v"(zb), MCX_R16(r));
}
// q = q + t_i, i = 0..N-1 in that order (t_i = lane i's t), as q = fma(t_i, 1.0, q): the same rounding of the same sum
#define MCX_GAD(i) "v_fmac_f64_dpp %0, %1, %2" MCX_DPPC(i)
#define MCX_GAD4 MCX_GAD(0) MCX_GAD(1) MCX_GAD(2) MCX_GAD(3)
#define MCX_GAD8 MCX_GAD4 MCX_GAD(4) MCX_GAD(5) MCX_GAD(6) MCX_GAD(7)
#define MCX_GAD12 MCX_GAD8 MCX_GAD(8) MCX_GAD(9) MCX_GAD(10) MCX_GAD(11)
#define MCX_GAD16 MCX_GAD12 MCX_GAD(12) MCX_GAD(13) MCX_GAD(14) MCX_GAD(15)
template <int N>
MCX_DEV void blk_addchain(double &q, double t)
{
    static_assert(N == 4 || N == 8 || N == 12 || N == 16, "blocks of four");
    const double one = 1.0;
    if constexpr (N == 4) asm("s_nop 1\n\t" MCX_GAD4 : "+v"(q) : "v"(t), "v"(one));
    else if constexpr (N == 8) asm("s_nop 1\n\t" MCX_GAD8 : "+v"(q) : "v"(t), "v"(one));
    else if constexpr (N == 12) asm("s_nop 1\n\t" MCX_GAD12 : "+v"(q) : "v"(t), "v"(one));
    else asm("s_nop 1\n\t" MCX_GAD16 : "+v"(q) : "v"(t), "v"(one));
}
// ss = fma(v_i, v_i, ss), i = 0..N-1 in that order (the banana target's and the data target's sums of squares)
#define MCX_GSQ(i) "v_mov_b64_dpp %2, %1" MCX_DPPC(i) "v_fmac_f64_dpp %0, %1, %2" MCX_DPPC(i)
#define MCX_GSQ4 MCX_GSQ(0) MCX_GSQ(1) MCX_GSQ(2) MCX_GSQ(3)
#define MCX_GSQ8 MCX_GSQ4 MCX_GSQ(4) MCX_GSQ(5) MCX_GSQ(6) MCX_GSQ(7)
#define MCX_GSQ12 MCX_GSQ8 MCX_GSQ(8) MCX_GSQ(9) MCX_GSQ(10) MCX_GSQ(11)
#define MCX_GSQ16 MCX_GSQ12 MCX_GSQ(12) MCX_GSQ(13) MCX_GSQ(14) MCX_GSQ(15)
template <int N>
MCX_DEV void blk_sqchain(double &ss, double v)
{
    static_assert(N == 4 || N == 8 || N == 12 || N == 16, "blocks of four");
    double tmp;
    if constexpr (N == 4) asm("s_nop 1\n\t" MCX_GSQ4 : "+v"(ss), "+v"(v), "=&v"(tmp));
    else if constexpr (N == 8) asm("s_nop 1\n\t" MCX_GSQ8 : "+v"(ss), "+v"(v), "=&v"(tmp));
    else if constexpr (N == 12) asm("s_nop 1\n\t" MCX_GSQ12 : "+v"(ss), "+v"(v), "=&v"(tmp));
    else asm("s_nop 1\n\t" MCX_GSQ16 : "+v"(ss), "+v"(v), "=&v"(tmp));
}

// the same four sequences for either group width: the asm statements above for rows of sixteen, the compiler's own DPP moves (quad_perm
// has no f64 form) and plain fmas for quads -- same operands, same order
template <int GW, int N>
MCX_DEV void gblk_fmac(double &p, double z, const double *r)
{
    if constexpr (GW == 16) blk_fmac<N>(p, z, r);
    else { static_assert(N == 4, "a quad"); p = dfma(r[0], grp_bcast<4, 0>(z), p); p = dfma(r[1], grp_bcast<4, 1>(z), p); p = dfma(r[2],
        grp_bcast<4, 2>(z), p); p = dfma(r[3], grp_bcast<4, 3>(z), p); }
}
template <int GW, int N>
MCX_DEV void gblk_fmac2(double &pa, double &pb, double za, double zb, const double *r)
{
    if constexpr (GW == 16) blk_fmac2<N>(pa, pb, za, zb, r);
    else {
        static_assert(N == 4, "a quad");
        pa = dfma(r[0], grp_bcast<4, 0>(za), pa); pb = dfma(r[0], grp_bcast<4, 0>(zb), pb);
        pa = dfma(r[1], grp_bcast<4, 1>(za), pa); pb = dfma(r[1], grp_bcast<4, 1>(zb), pb);
        pa = dfma(r[2], grp_bcast<4, 2>(za), pa); pb = dfma(r[2], grp_bcast<4, 2>(zb), pb);
        pa = dfma(r[3], grp_bcast<4, 3>(za), pa); pb = dfma(r[3], grp_bcast<4, 3>(zb), pb);
    }
}
template <int GW, int N>
MCX_DEV void gblk_addchain(double &q, double t)
{
    if constexpr (GW == 16) blk_addchain<N>(q, t);
    else { static_assert(N == 4, "a quad"); q = q + grp_bcast<4, 0>(t); q = q + grp_bcast<4, 1>(t); q = q + grp_bcast<4, 2>(t);
        q = q + grp_bcast<4, 3>(t); }
}
template <int GW, int N>
MCX_DEV void gblk_sqchain(double &ss, double v)
{
    if constexpr (GW == 16) blk_sqchain<N>(ss, v);
    else {
        static_assert(N == 4, "a quad");
        const double v0 = grp_bcast<4, 0>(v), v1 = grp_bcast<4, 1>(v), v2 = grp_bcast<4, 2>(v), v3 = grp_bcast<4, 3>(v);
        ss = dfma(v0, v0, ss); ss = dfma(v1, v1, ss); ss = dfma(v2, v2, ss); ss = dfma(v3, v3, ss);
    }
}

// ---------------------------------------------------------------- shapes
template <int D4, int GW = 16>
struct GDims {
    static constexpr int NS = (D4 + GW - 1) / GW;                                      // slots (columns / rows / positions) per lane
    static constexpr int rows(int s) { return D4 < GW * (s + 1) ? D4 : GW * (s + 1); }   // rows of the columns of slot s (upper triangle)
    static constexpr int off(int s) { int o = 0; for (int q = 0; q < s; ++q) o += rows(q); return o; }
    static constexpr int NR = off(NS);                                                 // doubles per lane for one triangular factor
    static constexpr int blk(int t) { return (D4 - GW * t) < GW ? (D4 - GW * t) : GW; }  // positions of block t
    static constexpr int ZS = D4 + 4;                                                  // LDS doubles per chain (normals + the saved slot)
    static constexpr int CPW = 64 / GW;                                                // chains per wave
    static constexpr unsigned GM = (GW == 16) ? 0xffffu : 0xfu;                        // a group's bits of a ballot
};

struct GChain {                         // the chain's stream: uniform over its 16 lanes ...
    uint64_t n; int saved; double saved_y;
    // ... and what the lane still holds of it: the Philox block `cb + l16` of the last round of attempts (cw), which very likely
    // contains the uniform MCMC_reject draws next (group_uniform)
    uint64_t cb; uint32_t cw[4];
};

#ifndef MCX_GROUP_SPLIT
// group_normals in two passes: attempts, then the logarithm / root / divisions for the kept pairs only (as gen_normals_split)
#define MCX_GROUP_SPLIT 1
#endif
// ---------------------------------------------------------------- normals: sixteen polar attempts of a chain at a time
// z[t] <- the chain's next npar deviates (position 16 t + l16; 0 in the padding), through the chain's LDS row.
template <int D4, int GW>
MCX_DEV void group_normals(uint32_t k0, uint32_t k1, GChain &g, double *zrow, int l16, int row, int d, bool act, double (&z)[GDims<D4,
    GW>::NS])
{
    using G = GDims<D4, GW>;
    int k = 0;
    MCX_WAVE_LDS_SYNC();                                         // the previous round's reads of the row are done
    // normal_bm's cached second deviate, mcmcrand.F90:172-175
    if (act && g.saved) { if (l16 == 0) zrow[0] = g.saved_y; g.saved = 0; k = 1; }
    bool need = act && (k < d);
    bool newsave = false;
    const int kst = k;                                           // where the drawn pairs start
    const bool drew = need;
    while (__any(need)) {
        const uint64_t b = (g.n >> 1) + (uint64_t)l16;
        const bool odd = (g.n & 1) != 0;
        uint32_t w0, w1, w2, w3;
        philox4x32_10((uint32_t)b, (uint32_t)(b >> 32), k0, k1, w0, w1, w2, w3);
        g.cb = g.n >> 1; g.cw[0] = w0; g.cw[1] = w1; g.cw[2] = w2; g.cw[3] = w3;
        // n odd (a single uniform was drawn since the last pair): attempt a takes the second half of block b and the first
        // half of block b + 1 = lane a + 1's block; the row's last lane has no such neighbour and sits the round out
        const uint32_t nw0 = (uint32_t)__shfl_down((int)w0, 1), nw1 = (uint32_t)__shfl_down((int)w1, 1);
        double x1 = odd ? bits_to_uniform(w2, w3) : bits_to_uniform(w0, w1);
        double x2 = odd ? bits_to_uniform(nw0, nw1) : bits_to_uniform(w2, w3);
        const bool valid = !(odd && l16 == GW - 1);
        x1 = 2.0 * x1 - 1.0; x2 = 2.0 * x2 - 1.0;
// tools/gen_bound.sh: every attempt accepted (NOT the reference's stream) -- the round count no pooling of attempts can beat
#ifdef MCX_PROBE_ALLOK
        if (!(x1 * x1 + x2 * x2 < 1.0)) { x1 *= 0.5; x2 *= 0.5; }
#endif
        const double xx = x1 * x1 + x2 * x2;
        const bool ok = valid && (xx < 1.0) && (xx != 0.0);
#if MCX_GROUP_SPLIT
        // parked unscaled where the deviates will stand; scaled below, the kept pairs only
        const double za = x2, zb = x1;
#else
        const double xs = ok ? xx : 0.5;
        const double zz = sqrt(-2.0 * d_log(xs) / xs);
        const double za = zz * x2, zb = zz * x1;                   // this call's deviate, the next call's (mcmcrand.F90:183-186)
#endif
        const uint32_t okm = (uint32_t)(__ballot(ok && need) >> (GW * row)) & G::GM;
        const int pre = __popc(okm & ((1u << l16) - 1u));          // accepted attempts before this one
        const int m = (d - k + 1) >> 1;                            // pairs the chain still needs
        const int tot = __popc(okm);
        const uint32_t mth = (uint32_t)(__ballot(ok && need && pre == m - 1) >> (GW * row)) & G::GM;
        if (ok && need && pre < m) {
            const int pos = k + 2 * pre;
            zrow[pos] = za;
            if (pos + 1 < d) zrow[pos + 1] = zb; else zrow[D4] = zb;
        }
        if (need) {
            if (tot >= m) {
                g.n += 2ull * (uint64_t)__ffs((int)mth);            // attempts up to and including the m-th accepted one
                newsave = ((d - k) & 1) != 0;                       // its second deviate is left over
                k = d; need = false;
            } else {
                g.n += odd ? 2ull * (GW - 1) : 2ull * GW;
                k += 2 * tot;
            }
        }
    }
    // (the wave's LDS operations retire in order: its reads below see its writes above)
    MCX_WAVE_LDS_SYNC();
#if MCX_GROUP_SPLIT
    // the second pass: z = sqrt(-2 log(xx) / xx) (mcmcrand.F90:183) for the pairs that were KEPT -- (npar + 1) / 2 of them, where the
    // attempts above were 16 (4) a round whether accepted, needed or neither -- pair p of the chain on lane p mod 16 (4); xx is formed
    // again from the same x1, x2 by the same two products and one sum
    {
        constexpr int NPB = ((D4 + 1) / 2 + GW - 1) / GW;
        const int mt = (d - kst + 1) >> 1;
        double px1[NPB], px2[NPB];
        sfor<0, NPB>([&](auto U) __attribute__((always_inline)) {
            constexpr int u = decltype(U)::value;
            const int pr = GW * u + l16, pos = kst + 2 * pr;
            const bool live = drew && pr < mt;
            px2[u] = zrow[live ? pos : 0];
            px1[u] = zrow[live ? (pos + 1 < d ? pos + 1 : D4) : 0];
        });
        sfor<0, NPB>([&](auto U) __attribute__((always_inline)) {
            constexpr int u = decltype(U)::value;
            const int pr = GW * u + l16, pos = kst + 2 * pr;
            const bool live = drew && pr < mt;
            const double xx0 = px1[u] * px1[u] + px2[u] * px2[u];
            const double xs = live ? xx0 : 0.5;
            const double zz = sqrt(-2.0 * d_log(xs) / xs);
            if (live) {
                zrow[pos] = zz * px2[u];
                zrow[pos + 1 < d ? pos + 1 : D4] = zz * px1[u];
            }
        });
        MCX_WAVE_LDS_SYNC();
    }
#endif
    sfor<0, G::NS>([&](auto T) __attribute__((always_inline)) {
        constexpr int t = decltype(T)::value;
        const int pos = GW * t + l16;
        z[t] = (pos < d) ? zrow[pos] : 0.0;
    });
    if (newsave) { g.saved = 1; g.saved_y = zrow[D4]; }
}

// one uniform for the chains with `take` (MCMC_reject, MCMC_DRAM.F90:150-151).  Uniform n sits in block n / 2; the lanes of the
// chain hold the blocks cb .. cb + 15 of its last round of polar attempts, and the attempts consumed there end right in front of n:
// the block is lane (n / 2 - cb)'s unless the round used up all sixteen -- then (any chain of the wave) every lane computes it.
template <int GW>
MCX_DEV double group_uniform(uint32_t k0, uint32_t k1, GChain &g, bool take, int row)
{
    const uint64_t blk = g.n >> 1;
    const uint64_t off = blk - g.cb;
    const bool have = off < (uint64_t)GW;
    const bool odd = (g.n & 1) != 0;
    const int src = (GW * row + (int)(off & (uint64_t)(GW - 1))) << 2;
    uint32_t lo = (uint32_t)__builtin_amdgcn_ds_bpermute(src, (int)(odd ? g.cw[2] : g.cw[0]));
    uint32_t hi = (uint32_t)__builtin_amdgcn_ds_bpermute(src, (int)(odd ? g.cw[3] : g.cw[1]));
    if (__any(take && !have)) {
        uint32_t x0, x1, x2, x3;
        philox4x32_10((uint32_t)blk, (uint32_t)(blk >> 32), k0, k1, x0, x1, x2, x3);
        if (!have) { lo = odd ? x2 : x0; hi = odd ? x3 : x1; }
    }
    if (take) g.n += 1;
    return bits_to_uniform(lo, hi);
}

// ---------------------------------------------------------------- checkbounds / priorfun / ssfunction on a group's vector
// (the per-parameter tables -- bounds, prior means and sigmas, the Gaussian target's mean -- are read where they are used: they are
//  shared by all chains and stay in the vector L1; holding them would cost ten registers per slot for the whole launch)
template <int D4, int GW>
MCX_DEV bool group_inbounds(const DevTarget &t, const double (&x)[GDims<D4, GW>::NS], int l16, int row, int d)
{
    using G = GDims<D4, GW>;
    if (!(t.lo || t.hi)) return true;
    bool bad = false;
    sfor<0, G::NS>([&](auto S) __attribute__((always_inline)) {
        constexpr int s = decltype(S)::value;
        const int c = GW * s + l16;
        if (c < d) {
            if (t.lo) bad = bad || !(x[s] > t.lo[c]);
            if (t.hi) bad = bad || !(x[s] < t.hi[c]);
        }
    });
    return ((uint32_t)(__ballot(bad) >> (GW * row)) & G::GM) == 0u;
}

// priorfun.f90:96-100: sum over the parameters with sigma > 0 of ((theta - mu) / sigma)**2, in index order from 0
template <int D4, int GW>
MCX_DEV double group_prior(const DevTarget &t, const double (&x)[GDims<D4, GW>::NS], int l16, int d)
{
    using G = GDims<D4, GW>;
    double p = 0.0;
    if (t.pmu) {
        sfor<0, G::NS>([&](auto S) __attribute__((always_inline)) {
            constexpr int s = decltype(S)::value;
            const int c = GW * s + l16;
            double qq = 0.0;
            if (c < d) { const double sg = t.psig[c]; if (sg > 0.0) { const double q = (x[s] - t.pmu[c]) / sg; qq = q * q; } }
            gblk_addchain<GW, G::blk(s)>(p, qq);          // (adding +0 for the parameters left out changes nothing: p >= +0)
        });
    }
    return p;
}

// TK: the target's kind at compile time (TGT_GAUSS / TGT_BANANA / TGT_EXPDATA), or -1 = whichever the engine holds: the three forms
// together cost a kernel ~100 registers more than its own form alone (the Gaussian one keeps sixteen matrix elements in flight)
template <int D4, int TK, int GW>
MCX_DEV double group_ss(const DevTarget &t, const double (&x)[GDims<D4, GW>::NS], int l16, int d, const double *laml)
{
    using G = GDims<D4, GW>;
    double ss = 0.0;
    // target_ss: ss = fma chain over theta_k**2, k ascending from 2
    if (TK == TGT_BANANA || (TK < 0 && t.kind == TGT_BANANA)) {
        const double th0 = grp_bcast<GW, 0>(x[0]), th1 = grp_bcast<GW, 1>(x[0]);
        const double t1 = th0 * th0;
        const double q = dfma(t.b, t1, th1) - 100.0 * t.b;
        ss = dfma(q, q, t1 / 100.0);
        sfor<0, G::NS>([&](auto S) __attribute__((always_inline)) {
            constexpr int s = decltype(S)::value;
            const double w = (s == 0 && l16 < 2) ? 0.0 : x[s];        // fma(0, 0, ss) = ss: positions 0, 1 and the padding drop out
            gblk_sqchain<GW, G::blk(s)>(ss, w);
        });
    // ss = fma chain over the residuals, data index ascending
    } else if (TK == TGT_EXPDATA || (TK < 0 && t.kind == TGT_EXPDATA)) {
        const double th0 = grp_bcast<GW, 0>(x[0]), th1 = grp_bcast<GW, 1>(x[0]);
        for (int base = 0; base < t.ndata; base += GW) {
            const int i = base + l16;
            double r = 0.0;
            if (i < t.ndata) r = t.y[i] - th0 * d_exp(-(th1 * t.x[i]));
            gblk_sqchain<GW, GW>(ss, r);
        }
    } else {                                          // Gaussian: mcxt_ss_gauss (oracle/mcx_targets.h), lane = row of Lam
        double v[G::NS], y[G::NS];
        sfor<0, G::NS>([&](auto S) __attribute__((always_inline)) { constexpr int s = decltype(S)::value; const int c = GW * s + l16;
            v[s] = (c < d) ? x[s] - t.mu[c] : 0.0; });
        sfor<0, G::NS>([&](auto S) __attribute__((always_inline)) {
            constexpr int s = decltype(S)::value;
            const int c = GW * s + l16;
            double ys = 0.0;
            sfor<0, G::NS>([&](auto TT) __attribute__((always_inline)) {
                constexpr int tb = decltype(TT)::value;
                constexpr int n = G::blk(tb);
                // row c of the precision matrix from the wave's LDS copy [j][c] (pitch D4, zero padding)
                double lam[n];
#pragma unroll
                for (int u = 0; u < n; ++u) lam[u] = laml[(GW * tb + u) * D4 + ((c < D4) ? c : D4 - 1)];
                gblk_fmac<GW, n>(ys, v[tb], lam);
                __builtin_amdgcn_sched_barrier(0);
            });
            y[s] = (c >= d) ? 0.0 : ys;
        });
        // per block of 16 rows, four partial chains q_k over the rows 16 b + k + 4 r (r = 0..3), summed into ss in the order b, k
        if constexpr (GW == 16) {
            sfor<0, G::NS>([&](auto S) __attribute__((always_inline)) {
                constexpr int s = decltype(S)::value;
                const int c = 16 * s + l16;
                // lane k < 4 collects lanes k + 4, k + 8, k + 12
                double q = y[s] * v[s];
                const double y4 = row_down<4>(y[s]), v4 = row_down<4>(v[s]), y8 = row_down<8>(y[s]), v8 = row_down<8>(v[s]),
                    y12 = row_down<12>(y[s]), v12 = row_down<12>(v[s]);
                if (c + 4 < d) q = dfma(y4, v4, q);
                if (c + 8 < d) q = dfma(y8, v8, q);
                if (c + 12 < d) q = dfma(y12, v12, q);
                const double q0 = row_bcast<0>(q), q1 = row_bcast<1>(q), q2 = row_bcast<2>(q), q3 = row_bcast<3>(q);
                if (16 * s + 0 < d) ss = (s == 0) ? q0 : ss + q0;
                if (16 * s + 1 < d) ss = ss + q1;
                if (16 * s + 2 < d) ss = ss + q2;
                if (16 * s + 3 < d) ss = ss + q3;
            });
        } else {
            // a quad: lane k owns the rows k, k + 4, ...: chain k of block b runs over the lane's own slots 4 b .. 4 b + 3
            sfor<0, (D4 + 15) / 16>([&](auto B) __attribute__((always_inline)) {
                constexpr int b = decltype(B)::value;
                double q = y[4 * b] * v[4 * b];
                sfor<1, 4>([&](auto R) __attribute__((always_inline)) {
                    constexpr int r = decltype(R)::value;
                    if constexpr (4 * b + r < G::NS) { if (16 * b + 4 * r + l16 < d) q = dfma(y[4 * b + r], v[4 * b + r], q); }
                });
                const double q0 = grp_bcast<4, 0>(q), q1 = grp_bcast<4, 1>(q), q2 = grp_bcast<4, 2>(q), q3 = grp_bcast<4, 3>(q);
                if (16 * b + 0 < d) ss = (b == 0) ? q0 : ss + q0;
                if (16 * b + 1 < d) ss = ss + q1;
                if (16 * b + 2 < d) ss = ss + q2;
                if (16 * b + 3 < d) ss = ss + q3;
            });
        }
    }
    return ss;
}

// ---------------------------------------------------------------- the kernel
// Iterations it0..it1 of MCMC_run (MCMC_run.F90:41-107) for the four chains of a wave: method = 'dram', per-chain Cholesky factors,
// one of the single-launch device targets, nycol = 1.  accb: one byte per chain and iteration of the launch
// (accepted or not); group_pack_kernel turns them into the tile ballots the adaptation and the chain decoder read.
// DRM = 0: no delayed rejection.  R in registers; two waves per SIMD.
// DRM = 1: delayed rejection, any drscale: R, R2 and iC in registers (accumulator registers included); one wave per SIMD.
// DRM = 2: delayed rejection with drscale a power of two: R2 = R / drscale is then exact element by element, and so is every
//          product and every partial sum of the second-stage proposal -- R2'z = (R'z) / drscale bit for bit as long as nothing
//          leaves the normal range, which holds when every nonzero |R(i,j)| lies in [2**-500, 2**500] (group_check_kernel looks
//          at every chain's factor whenever an adaptation has rewritten it and raises a device flag otherwise; the host queues
//          this kernel AND the DRM = 1 one behind that flag, and the one it does not select returns at once).  R in registers,
//          iC as a full symmetric square in LDS (the quadratic forms read it with immediate offsets); two waves per SIMD up to
//          npar = 24 (quads: up to npar 12 -- at 13..16 the allocator cannot meet two, the kernel declares one and takes the registers).
#ifndef MCX_GROUP_WAVES2
#define MCX_GROUP_WAVES2 2
#endif
#ifndef MCX_GROUP_GAUSS1
// the Gaussian target from this D4 on: one wave per SIMD (its matrix-vector product spills at 256 registers)
#define MCX_GROUP_GAUSS1 99
#endif
#ifndef MCX_GROUP_WAVES4
#define MCX_GROUP_WAVES4 2       // quads (GW = 4)
#endif
template <int GW, int D4, int DRM, int TK>
__global__ __launch_bounds__(64,
    (DRM == 1 || D4 > 32 || (DRM == 2 && GW == 4 && D4 > 12) || (TK == TGT_GAUSS && D4 >= MCX_GROUP_GAUSS1)) ? 1 : (GW == 4
    ? MCX_GROUP_WAVES4 : MCX_GROUP_WAVES2)) void group_step_kernel(EngineDev E, int it0, int it1, const double *__restrict__ g_lamT,
    uint8_t *accb,
                                                                                                   const int *__restrict__ gflag, int want)
{
    using G = GDims<D4, GW>;
    constexpr int NS = G::NS, CPW = G::CPW;
    constexpr bool DR = DRM != 0;
    // the other instantiation has this launch (wave-uniform: every wave reads the same word)
    if (gflag && ((*gflag != 0) != (want != 0))) return;
    // LDS: [iC squares of the four chains (DRM = 2)] [the chains' normal rows]: a padding lane's read past its chain's square lands in the
    // next square or in the normal rows (finite or not, its result is discarded)
    constexpr int SQ = (DRM == 2) ? D4 * D4 : 0;
    // ... and (two waves per SIMD: registers are short) the columns of R's LAST, partly filled slot -- npar - 16 (NS - 1) columns of
    // which only as many lanes hold anything: [row][column] per chain, read back with immediate offsets
    constexpr int LS = (DRM != 1 && NS > 1 && (D4 % GW) != 0) ? NS - 1 : -1;      // the slot kept in LDS (-1: none)
    constexpr int LC = (LS >= 0) ? D4 - GW * LS : 0;                              // its columns
    constexpr int RL = LC * D4;                                                   // doubles per chain
    constexpr int NRR = (LS >= 0) ? G::off(LS) : G::NR;                           // doubles of R per lane that stay in registers
    // the Gaussian target's precision matrix, [j][i] with pitch D4
    constexpr int LQ = (TK == TGT_GAUSS || TK < 0) ? D4 * D4 : 0;
    __shared__ double lds[CPW * SQ + CPW * RL + CPW * G::ZS + LQ];
    // (l16: the lane's place in its group, row: the group = chain of the wave)
    const int lane = threadIdx.x, l16 = lane & (GW - 1), row = lane / GW, d = E.d;
    const int chain = blockIdx.x * CPW + row, tile = chain >> 6, cl = chain & 63;
    const size_t nslots = (size_t)E.ntiles * 64;
    double *zrow = lds + CPW * SQ + CPW * RL + row * G::ZS;
    const double *icl = lds + row * SQ;
    const double *laml = lds + CPW * SQ + CPW * RL + CPW * G::ZS;
    if constexpr (LQ > 0) {
        if (E.tgt.kind == TGT_GAUSS) {
            for (int e = lane; e < D4 * D4; e += 64) { const int j = e / D4, i = e % D4;
                lds[CPW * SQ + CPW * RL + CPW * G::ZS + e] = (i < d && j < d) ? g_lamT[(size_t)j * d + i] : 0.0; }
        }
    }
    // (lanes past the slot's columns read its last one: finite, discarded)
    const double *rll = lds + CPW * SQ + row * RL + ((l16 < LC) ? l16 : (LC > 0 ? LC - 1 : 0));

    // ---- factors into registers / LDS (once per launch)
    double Rr[NRR > 0 ? NRR : 1], R2r[DRM == 1 ? G::NR : 1], Sr[DRM == 1 ? NS * D4 : 1];
    {
        const double *Rt = E.R + (size_t)tile * E.P * 64;
        const double *R2t = DR ? E.R2 + (size_t)tile * E.P * 64 : nullptr;
        const double *iCt = DR ? E.iC + (size_t)tile * E.P * 64 : nullptr;
        sfor<0, NS>([&](auto S) __attribute__((always_inline)) {
            constexpr int s = decltype(S)::value;
            const int c = GW * s + l16;
#pragma unroll
            for (int i = 0; i < G::rows(s); ++i) {
                const bool in = (c < d) && (i <= c);
                const size_t e = in ? (size_t)pidx(i, c, d) : 0;
                const double r = Rt[e * 64 + cl];
                if constexpr (s == LS) { if (l16 < LC) lds[CPW * SQ + row * RL + i * LC + l16] = in ? r : 0.0; }
                else Rr[G::off(s) + i] = in ? r : 0.0;
                if constexpr (DRM == 1) { const double r2 = R2t[e * 64 + cl]; R2r[G::off(s) + i] = in ? r2 : 0.0; }
            }
            if constexpr (DRM == 1) {
#pragma unroll
                for (int j = 0; j < D4; ++j) {
                    const bool in = (c < d) && (j < d);
                    const size_t e = in ? (size_t)((c <= j) ? pidx(c, j, d) : pidx(j, c, d)) : 0;
                    const double v = iCt[e * 64 + cl];
                    Sr[s * D4 + j] = in ? v : 0.0;
                }
            }
            if constexpr (DRM == 2) {
                if (c < D4) {
#pragma unroll 4
                    for (int j = 0; j < D4; ++j) {
                        const bool in = (c < d) && (j < d);
                        const size_t e = in ? (size_t)((c <= j) ? pidx(c, j, d) : pidx(j, c, d)) : 0;
                        const double v = iCt[e * 64 + cl];
                        lds[row * SQ + j * D4 + c] = in ? v : 0.0;
                    }
                }
            }
        });
    }
    // the fills above (Lam, R's last slot, iC) before any lane reads another lane's part
    MCX_WAVE_LDS_SYNC();
    const double inv2 = (DRM == 2) ? 1.0 / E.drscale : 1.0;   // exact: drscale is a power of two in this instantiation
    double th[NS];
    sfor<0, NS>([&](auto S) __attribute__((always_inline)) {
        constexpr int s = decltype(S)::value;
        const int c = GW * s + l16, cc = c < d ? c : 0;
        th[s] = (c < d) ? TIDX(E.theta, tile, d, cc, cl) : 0.0;
    });
    GChain g;
    g.n = TIDX(E.rngn, tile, 1, 0, cl);
    g.saved = (int)TIDX(E.ictr, tile, NICTR, I_SAVED, cl);
    g.saved_y = TIDX(E.scal, tile, NSCAL, S_SAVEDY, cl);
    g.cb = (g.n >> 1) + (1ull << 62); g.cw[0] = g.cw[1] = g.cw[2] = g.cw[3] = 0u;      // nothing held yet: n / 2 - cb is far from 0..15
    const uint32_t k0 = E.k0, k1 = E.chain_id0 + (uint32_t)chain;
    double ss1 = TIDX(E.scal, tile, NSCAL, S_SS1, cl), pri1 = TIDX(E.scal, tile, NSCAL, S_PRI1, cl);
    double sigma2 = TIDX(E.scal, tile, NSCAL, S_SIGMA2, cl);
    double alpha12 = TIDX(E.scal, tile, NSCAL, S_ALPHA12, cl);
    uint32_t stayed = TIDX(E.ictr, tile, NICTR, I_STAYED, cl), bnd = TIDX(E.ictr, tile, NICTR, I_BNDSTAYED, cl);
    uint32_t chainind = TIDX(E.ictr, tile, NICTR, I_CHAININD, cl), curcount = TIDX(E.ictr, tile, NICTR, I_CURCOUNT, cl);
    uint32_t dracc = TIDX(E.ictr, tile, NICTR, I_DRACC, cl), drtries = TIDX(E.ictr, tile, NICTR, I_DRTRIES, cl);
    uint32_t erstayed = TIDX(E.ictr, tile, NICTR, I_ERSTAYED, cl);
    const bool er = !DR && E.method == M_ER;           // MCMC_run_er: the threshold is drawn before ss is looked at

    // One pass of the loop below is one STAGE of every chain of the wave -- the first stage of its iteration, or (delayed rejection) the
    // second one when the first was rejected -- so the four chains of a wave drift apart by whole iterations instead of all of
    // them sitting through a second stage whenever one of them needs it (MCMC_run.F90:41-107: 1.48 stages per iteration at
    // BASELINE config 3's acceptance, against two in lockstep).  Both stages are the same code: normals, proposal with R (times
    // 1 / drscale, or with R2), bounds / prior / ss of the candidate; then MCMC_alpha for a first stage, MCMC_DR_alpha13 for a
    // second, MCMC_reject for both.  A chain's own sequence of draws and operations is untouched.
    int it = it0;                                       // the chain's iteration (uniform over its 16 lanes)
    bool st2 = false;                                   // ... and whether it is at its second stage
    double c1[NS], ss2s = 0.0, pri2s = 0.0;             // the rejected first-stage candidate with its ss and prior
    sfor<0, NS>([&](auto S) __attribute__((always_inline)) { c1[decltype(S)::value] = 0.0; });
    // -DMCX_PHASE_PROF (tools/build_variant.sh; profiles/r05_e/c3_phases.txt): where a pass of the loop goes, by wall_clock64
#ifdef MCX_PHASE_PROF
    unsigned long long gph[8] = {0, 0, 0, 0, 0, 0, 0, 0}, gtq = wall_clock64(), gpasses = 0;
#define GPH(i) { unsigned long long tn = wall_clock64(); gph[i] += tn - gtq; gtq = tn; }
#else
#define GPH(i)
#endif
    while (__any(it <= it1)) {
        const bool act = it <= it1;
        // ---- newpar = MCMC_propose(oldpar, R) / newpar2 = MCMC_propose(oldpar, R2): the stage's first draws
        double z[NS], cand[NS];
        group_normals<D4, GW>(k0, k1, g, zrow, l16, row, d, act, z);
        GPH(0)
        const bool any2 = DR && __any(act && st2);
        sfor<0, NS>([&](auto S) __attribute__((always_inline)) {
            constexpr int s = decltype(S)::value;
            double p = 0.0;
            sfor<0, (G::rows(s) + GW - 1) / GW>([&](auto TT) __attribute__((always_inline)) {
                constexpr int tb = decltype(TT)::value;
                constexpr int n = (G::rows(s) - GW * tb) < GW ? (G::rows(s) - GW * tb) : GW;
                if constexpr (s == LS) {
                    double r[n];
#pragma unroll
                    for (int u = 0; u < n; ++u) r[u] = rll[(GW * tb + u) * LC];
                    gblk_fmac<GW, n>(p, z[tb], r);
                    __builtin_amdgcn_sched_barrier(0);
                } else gblk_fmac<GW, n>(p, z[tb], &Rr[(s == LS) ? 0 : G::off(s) + GW * tb]);
            });
            if constexpr (s == LS) p = (l16 < LC) ? p : 0.0;         // (the padding lanes multiplied somebody else's column)
            if constexpr (DRM == 2) p = st2 ? p * inv2 : p;          // exact (see above)
            if constexpr (DRM == 1) {
                if (any2) {
                    double p2 = 0.0;
                    sfor<0, (G::rows(s) + GW - 1) / GW>([&](auto TT) __attribute__((always_inline)) {
                        constexpr int tb = decltype(TT)::value;
                        constexpr int n = (G::rows(s) - GW * tb) < GW ? (G::rows(s) - GW * tb) : GW;
                        gblk_fmac<GW, n>(p2, z[tb], &R2r[(DRM == 1) ? G::off(s) + GW * tb : 0]);
                    });
                    p = st2 ? p2 : p;
                }
            }
            cand[s] = th[s] + p;                         // newpar = oldpar + R'z
        });
        GPH(1)
        // ---- bounds, prior, ss of the candidate
        const bool inb = group_inbounds<D4, GW>(E.tgt, cand, l16, row, d);
        const double pri = group_prior<D4, GW>(E.tgt, cand, l16, d);
        const double ss = group_ss<D4, TK, GW>(E.tgt, cand, l16, d, laml);
        GPH(2)
        // ---- second stages: MCMC_DR_alpha13's two quadratic forms dx' iC dx, dx_a = newpar2 - newpar, dx_b = oldpar - newpar
        // (MCMC_DRAM.F90:176-182)
        double qa = 0.0, qb = 0.0;
        if constexpr (DR) {
            if (any2) {
                double xa[NS], xb[NS], ta[NS], tb_[NS];
                sfor<0, NS>([&](auto S) __attribute__((always_inline)) { constexpr int s = decltype(S)::value; xa[s] = cand[s] - c1[s];
                    xb[s] = th[s] - c1[s]; });
                sfor<0, NS>([&](auto S) __attribute__((always_inline)) {
                    constexpr int s = decltype(S)::value;
                    double ya = -0.0, yb = -0.0;                      // first term a plain product (see blk_fmac2)
                    sfor<0, NS>([&](auto TT) __attribute__((always_inline)) {
                        constexpr int tb = decltype(TT)::value;
                        constexpr int n = G::blk(tb);
                        if constexpr (DRM == 1) gblk_fmac2<GW, n>(ya, yb, xa[tb], xb[tb], &Sr[(DRM == 1) ? s * D4 + GW * tb : 0]);
                        else {
                            double r[n];
#pragma unroll
                            for (int u = 0; u < n; ++u) r[u] = icl[(GW * tb + u) * D4 + GW * s + l16];
                            gblk_fmac2<GW, n>(ya, yb, xa[tb], xb[tb], r);
                            __builtin_amdgcn_sched_barrier(0);          // (keeps the next block's LDS reads from piling up in registers)
                        }
                    });
                    const bool in = GW * s + l16 < d;
                    ta[s] = in ? ya * xa[s] : 0.0; tb_[s] = in ? yb * xb[s] : 0.0;
                });
                sfor<0, NS>([&](auto S) __attribute__((always_inline)) { constexpr int s = decltype(S)::value; gblk_addchain<GW,
                    G::blk(s)>(qa, ta[s]); gblk_addchain<GW, G::blk(s)>(qb, tb_[s]); });
            }
        }
        GPH(3)
        // ---- MCMC_alpha (MCMC_DRAM.F90:100-118) or MCMC_DR_alpha13 (:162-186), then MCMC_reject (:140-155)
        bool rej = true, take = false, fin = false;
        double alpha = 0.0;
        {
            // tst of MCMC_alpha for a first stage, l2 of MCMC_DR_alpha13 for a second: the same expression of the same operands
            // (x / 1.0 is x: with sigma2 = 1 on every chain of the wave -- no sigma2 update, the reference's default -- the divisions are
            // skipped)
            const bool s2one = __all(sigma2 == 1.0);
            double dss = ss - ss1, dss2 = ss2s - ss;
            if (!s2one) { dss = dss / sigma2; dss2 = dss2 / sigma2; }
            const double tstl = -0.5 * (dss + (pri - pri1));
            // ONE exponential for both kinds of chains: exp(tst) (MCMC_alpha) or the alpha32 term (MCMC_DR_alpha13); every lane evaluates
            // it whether its chain needs the value or not (d_exp has no side effects), which takes two divergent calls out of the pass
            const double a32 = -0.5 * (dss2 + (pri2s - pri));
            const double e1 = d_exp((DR && st2) ? a32 : tstl);
            if (act) {
                // early rejection, MCMC_run_er.F90:60-89: u is always drawn (MCMC_sscrit)
                if (er) {
                    if (!inb) bnd += 1;
                    else take = true;
                } else if (!(DR && st2)) {
                    // MCMC_run.F90:49 (with DR an out-of-bounds first stage is not counted)
                    if (!inb) { alpha12 = 0.0; if (!DR) bnd += 1; }
                    else {
                        alpha12 = (tstl >= 0.0) ? 1.0 : ((tstl < -708.39641853226408) ? 0.0 : e1);      // d_alpha
                        if (alpha12 >= 1.0) rej = false;
                        else if (alpha12 > 0.0) take = true;
                    }
                    alpha = alpha12;
                } else {
                    if (!inb) bnd += 1;
                    else {
                        const double alpha32 = (alpha12 == 0.0) ? 0.0 : min1(e1);
                        const double q1 = -0.5 * (qa - qb);
                        alpha = min1(d_exp(tstl + q1) * (1.0 - alpha32) / (1.0 - alpha12));
                        if (alpha >= 1.0) rej = false;
                        else if (alpha > 0.0) take = true;
                    }
                }
            }
        }
        GPH(4)
        if (__any(take)) {
            const double u = group_uniform<GW>(k0, k1, g, take, row);
            if (er) {
                // sscrit = -2 log u + ss1 / sigma2 + pri1 (MCMC_DRAM.F90:124-135)
                if (take) {
                    double sscrit = -2.0 * d_log(u) + ss1 / sigma2 + pri1;
                    if (pri >= sscrit) erstayed += 1;                        // rejected on the prior alone
                    else { sscrit = sigma2 * (sscrit - pri); rej = (ss >= sscrit); }
                }
            } else if (take && u <= alpha) rej = false;
        }
        GPH(5)
        if (act) {
            if (DR && !st2 && rej) {                     // on to the second stage: one delayed-rejection try (MCMC_run.F90:65-91)
                st2 = true; drtries += 1; ss2s = ss; pri2s = pri;
                sfor<0, NS>([&](auto S) __attribute__((always_inline)) { constexpr int s = decltype(S)::value; c1[s] = cand[s]; });
            } else {                                     // the iteration ends with this stage
                if (rej) { stayed += 1; curcount += 1; }
                else {
                    if (DR && st2) dracc += 1;
                    ss1 = ss; pri1 = pri; chainind += 1; curcount = 1;
                    // oldpar = newpar; MCMC_savechain (MCMC_aux.F90:167-185): accepted row into the ring
                    sfor<0, NS>([&](auto S) __attribute__((always_inline)) { constexpr int s = decltype(S)::value; th[s] = cand[s]; });
                    if (E.hist) {
                        double *h = E.hist + ((size_t)tile * E.wcap + (it % E.wcap)) * (size_t)E.hs * 64;
                        sfor<0, NS>([&](auto S) __attribute__((always_inline)) { constexpr int s = decltype(S)::value;
                            const int c = GW * s + l16; if (c < d) h[(size_t)c * 64 + cl] = th[s]; });
                        if (l16 == 0) h[(size_t)d * 64 + cl] = ss1;
                    }
                }
                // the accept byte for the ballots
                if (accb && l16 == 0) accb[(size_t)(it - it0) * nslots + chain] = rej ? (uint8_t)0 : (uint8_t)1;
                fin = true;
            }
        }
        // ---- MCMC_updatesigma2 (MCMC_DRAM.F90:192-206) of the chains whose iteration ended: random_gamma's rejection loops draw a
        // data-dependent number of deviates and uniforms one after the other (mcmcrand.F90:86-162), so every lane of the chain runs the
        // chain's own sampler on the chain's stream -- sixteen identical copies; the polar cache changes hands through it as in the
        // reference.  (A wave pays one sampler for four chains here where the lane kernels pay one for 64: few chains, not many.)
        if (E.updatesigma && __any(fin)) {
            if (fin) {
                Rng q;
                q.k0 = k0; q.k1 = k1; q.n = g.n; q.cblk = 0; q.c2 = 0; q.c3 = 0; q.saved = g.saved; q.saved_y = g.saved_y;
                const double gm = rng_gamma(q, E.gam_shape, 2.0 / (E.N0S02 + ss1));
                sigma2 = 1.0 / gm;
                g.n = q.n; g.saved = q.saved; g.saved_y = q.saved_y;
            }
        }
        if (fin) {
            if (E.hist && E.record_s2 && l16 == 0) E.s2hist[((size_t)tile * E.wcap + (it % E.wcap)) * 64 + cl] = sigma2;
            it += 1; st2 = false;
        }
        GPH(6)
#ifdef MCX_PHASE_PROF
        gpasses += 1;
#endif
    }
#ifdef MCX_PHASE_PROF
    if (lane == 0 && (blockIdx.x == 0 || blockIdx.x == gridDim.x / 2 || blockIdx.x == gridDim.x - 1))
        printf("group_step block %d: %llu passes for %d iterations, x10ns: normals %llu proposal %llu bounds+prior+ss %llu DR quadratic forms %llu alpha(exp) %llu uniform+decide %llu bookkeeping+history %llu\n",
               (int)blockIdx.x, gpasses, it1 - it0 + 1, gph[0], gph[1], gph[2], gph[3], gph[4], gph[5], gph[6]);
#endif
#undef GPH

    sfor<0, NS>([&](auto S) __attribute__((always_inline)) { constexpr int s = decltype(S)::value; const int c = GW * s + l16;
        if (c < d) TIDX(E.theta, tile, d, c, cl) = th[s]; });
    if (l16 == 0) {
        TIDX(E.scal, tile, NSCAL, S_SIGMA2, cl) = sigma2;
        TIDX(E.rngn, tile, 1, 0, cl) = g.n;
        TIDX(E.ictr, tile, NICTR, I_SAVED, cl) = (uint32_t)g.saved;
        TIDX(E.scal, tile, NSCAL, S_SAVEDY, cl) = g.saved_y;
        TIDX(E.scal, tile, NSCAL, S_SS1, cl) = ss1; TIDX(E.scal, tile, NSCAL, S_PRI1, cl) = pri1;
        TIDX(E.scal, tile, NSCAL, S_ALPHA12, cl) = alpha12;
        TIDX(E.ictr, tile, NICTR, I_STAYED, cl) = stayed; TIDX(E.ictr, tile, NICTR, I_BNDSTAYED, cl) = bnd;
        TIDX(E.ictr, tile, NICTR, I_CHAININD, cl) = chainind; TIDX(E.ictr, tile, NICTR, I_CURCOUNT, cl) = curcount;
        TIDX(E.ictr, tile, NICTR, I_DRACC, cl) = dracc; TIDX(E.ictr, tile, NICTR, I_DRTRIES, cl) = drtries;
        TIDX(E.ictr, tile, NICTR, I_ERSTAYED, cl) = erstayed;
    }
}

// DRM = 2's precondition: *flag |= 1 when some chain's factor holds a nonzero element outside [2**-500, 2**500] (lane = chain)
__global__ void group_check_kernel(EngineDev E, int *flag)
{
    const int lane = threadIdx.x, tile = blockIdx.x;
    const double *Rt = E.R + (size_t)tile * E.P * 64;
    bool bad = false;
    for (int e = 0; e < E.P; ++e) { const double r = GV(Rt, e); const double ar = fabs(r);
        bad = bad || (r != 0.0 && !(ar >= 0x1.0p-500 && ar <= 0x1.0p500)); }
    if (__any(bad) && lane == 0) atomicOr(flag, 1);
}

// ---------------------------------------------------------------- MCMC_calculate_R with the matrices in LDS (Cholesky branch,
// MCMC_adapt.F90:211-225)
// adapt_post_kernel's factorisation for the chains whose tick recomputes the factor (ADF_DOCALC), any chain count, npar <= 64: a WORKGROUP
// of NW
// waves owns 4 NW neighbouring chains of a tile and moves their matrices between HBM and LDS cooperatively, 32 NW contiguous bytes per
// element --
// whole 128-byte lines at NW = 4 -- and the workgroups of one tile sit on one XCD (blockIdx round-robins over the eight), whose L2 combines
// what
// is left.  The LDS copy is the PACKED upper triangle by rows (every operand of dpotf2 / dtrti2 / dlauu2 lies in it), so that npar 50 --
// BASELINE
// config 4's size with method = 'dram' -- holds four chains in 41 kB and three workgroups share a CU.  Sixteen lanes per chain, a lane owns
// NC = ceil(npar / 16) columns (dpotf2) / rows (dtrti2, dlauu2); every chain of operations is the one of calculate_R / potri_packed
// (mcx_adapt.hpp):
// dpotf2('U'):   lane = COLUMN k of the factor; step j adds T(i,j) T(i,k), i < j ascending, to column k's chain (T(i,j) is a broadcast
// read),
//                  every lane also runs the pivot's own chain, so the pivot needs no exchange; R = T 2.4 / sqrt(npar); R2 = R / drscale;
// dtrti2('U'):   lane = ROW r of the inverse; element (r,j) = -1/A(j,j) x [A(r,j) Ainv(r,r), then + A(jj,j) Ainv(r,jj) for jj = r+1..j-1
// ascending]:
//                  dtrmv's column sweep read per ROW -- the temp of sweep jj is the ORIGINAL A(jj,j), sweep r starts row r's chain with the
//                  product, the sweeps behind it add their terms in order -- so the rows of one column are independent chains;
//   dlauu2('U'):   lane = ROW r: A(r,i) = A(i,i) A(r,i) + sum_{k>i} A(i,k) A(r,k) ascending k; the diagonal's sum of squares on every lane.
// adapt_post_kernel streamed the packed matrices of a tile ~5.6 times through L2 / HBM for its 8 x 8 register blocks (33.5 ms per tick of
// 1 048 576 chains at npar 50, 2.2 ms at config 3's size); this reads and writes each once.  Results: R, R2, iC, I_INFO and the status
// bits, as adapt_post_kernel leaves them.  (Round 4's one-wave form with [row][column] squares in LDS, group_factor_kernel, is superseded:
// tools/variants/README.md.) Grid: 8 ceil(ntiles / 8) (64 / (4 NW)) workgroups of 64 NW threads; LDS: 4 NW (P | 1) doubles + 4 NW ints.
template <int NC, int NW>
__global__ __launch_bounds__(64 * NW) void tile_factor_kernel(EngineDev E)
{
    extern __shared__ double Mf[];
    // chains per workgroup, workgroups per tile, elements per cooperative pass
    constexpr int CH = 4 * NW, WPT = 64 / CH, ES = 64 * NW / CH;
    const int wg = blockIdx.x, xcd = wg & 7, idx = wg >> 3;
    const int tile = (idx / WPT) * 8 + xcd, part = idx % WPT;
    if (tile >= E.ntiles) return;                                          // (uniform over the workgroup)
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, l16 = lane & 15, row = lane >> 4;
    const int d = E.d, P = E.P, PS = P | 1;
    const int cl0 = part * CH, cl = cl0 + 4 * w + row;                      // this lane's chain of the tile
    // its packed upper triangle: element (i,k), i <= k, at i d - i (i + 1) / 2 + k
    double *M = Mf + (size_t)(4 * w + row) * PS;
    int *flg = (int *)(Mf + (size_t)CH * PS);
    const bool act = (TIDX(E.ictr, tile, NICTR, I_ADFLAGS, cl) & ADF_DOCALC) != 0;
    if (!__syncthreads_or(act ? 1 : 0)) return;
    // cooperative moves: chain cc of the workgroup, elements e0, e0 + ES, ...
    const int cc = tid % CH, e0 = tid / CH;
    // (sixteen elements' loads before their LDS stores: element by element every load is a round trip of its own -- 80 in a row at npar 50)
    constexpr int CB = 16;
    {
        const double *Cg = E.cmat + (size_t)tile * P * 64 + cl0 + cc;
        double *Mc = Mf + (size_t)cc * PS;
        for (int e = e0; e < P; e += CB * ES) {
            double v[CB];
#pragma unroll
            for (int u = 0; u < CB; ++u) { const int ee = e + u * ES; v[u] = Cg[(size_t)(ee < P ? ee : P - 1) * 64]; }
#pragma unroll
            for (int u = 0; u < CB; ++u) { const int ee = e + u * ES; if (ee < P) Mc[ee] = v[u]; }
        }
    }
    __syncthreads();
    // this lane's columns / rows, and clamped for the branch-free loops
    int kk[NC], kc[NC];
#pragma unroll
    for (int q = 0; q < NC; ++q) { kk[q] = l16 + 16 * q; kc[q] = kk[q] < d ? kk[q] : d - 1; }
    // ---- dpotf2('U'): lane = column k; step j adds T(i,j) T(i,k), i < j ascending, to column k's chain; every lane runs the pivot's own
    // chain
    int info = 0;
    for (int j = 0; j < d; ++j) {
        MCX_WAVE_LDS_SYNC();                                                // row j - 1 (other lanes' columns) is written
        double acc[NC], accj = 0.0;
#pragma unroll
        for (int q = 0; q < NC; ++q) acc[q] = 0.0;
        // rowstart(i) - i: element (i,k) at rb + k  (k < i: in bounds, dropped)
        int rb = 0;
#pragma unroll 4
        for (int i = 0; i < j; ++i) {
            const double tij = M[rb + j];
            double m[NC];
#pragma unroll
            for (int q = 0; q < NC; ++q) m[q] = M[rb + kc[q]];
            accj = dfma(tij, tij, accj);
#pragma unroll
            for (int q = 0; q < NC; ++q) acc[q] = dfma(tij, m[q], acc[q]);
            rb += d - (i + 1);
        }
        const double ajj = M[rb + j] - accj;
        if (act && info == 0 && !(ajj > 0.0)) info = j + 1;
        const double rjj = sqrt(ajj), rinv = 1.0 / rjj;
        double nv[NC];
#pragma unroll
        for (int q = 0; q < NC; ++q) nv[q] = (M[rb + kc[q]] - acc[q]) * rinv;
#pragma unroll
        for (int q = 0; q < NC; ++q) {
            if (kk[q] == j) M[rb + j] = rjj;
            if (kk[q] > j && kk[q] < d) M[rb + kk[q]] = nv[q];
        }
    }
    const bool ok = act && info == 0;
    if (act && l16 == 0) {
        TIDX(E.ictr, tile, NICTR, I_INFO, cl) = (uint32_t)info;
        if (info != 0) TIDX(E.ictr, tile, NICTR, I_STATUS, cl) |= ST_CHOL_FAIL;          // warning, old R kept (MCMC_adapt.F90:168-171)
    }
    MCX_WAVE_LDS_SYNC();
    // ---- R = T 2.4 / sqrt(npar), out with it -- and R2 = R / drscale -- cooperatively; the LDS copy takes the scaling too (dpotri works
    // on R)
    if (l16 == 0) flg[4 * w + row] = ok ? 1 : 0;
    __syncthreads();
    if (flg[cc]) {
        const double sq = sqrt((double)d);
        double *Rg = E.R + (size_t)tile * P * 64 + cl0 + cc;
        double *R2g = E.dodr ? E.R2 + (size_t)tile * P * 64 + cl0 + cc : nullptr;
        double *Mc = Mf + (size_t)cc * PS;
        for (int e = e0; e < P; e += CB * ES) {
            double v[CB];
#pragma unroll
            for (int u = 0; u < CB; ++u) { const int ee = e + u * ES; v[u] = Mc[ee < P ? ee : P - 1] * 2.4 / sq; }
#pragma unroll
            for (int u = 0; u < CB; ++u) {
                const int ee = e + u * ES;
                if (ee < P) { Rg[(size_t)ee * 64] = v[u]; if (R2g) { R2g[(size_t)ee * 64] = v[u] / E.drscale; Mc[ee] = v[u]; } }
            }
        }
    }
    if (!E.dodr) return;
    // (also: the scaled copies above are written before dpotri reads them)
    if (!__syncthreads_or(ok ? 1 : 0)) return;
    // ---- iC = dpotri('U', R): dtrti2 then dlauu2, in place on the scaled factor; lane = ROW r
    int info2 = 0;
    { int rb = 0; for (int j = 0; j < d; ++j) { if (ok && info2 == 0 && M[rb + j] == 0.0) info2 = j + 1; rb += d - (j + 1); } }
    const bool go = ok && info2 == 0;
    if (ok && info2 != 0 && l16 == 0) TIDX(E.ictr, tile, NICTR, I_STATUS, cl) |= ST_POTRI_FAIL;                   // the reference stops
    int rbr[NC];                                                            // rowstart(r) - r of this lane's (clamped) rows
#pragma unroll
    for (int q = 0; q < NC; ++q) rbr[q] = kc[q] * d - kc[q] * (kc[q] + 1) / 2;
    if (__any(go)) {
        int rbj = 0;                                                        // row j
        // dtrti2: element (r,j) = -1/A(j,j) x [A(r,j) Ainv(r,r), then + A(jj,j) Ainv(r,jj), jj = r+1..j-1 ascending]
        for (int j = 0; j < d; ++j) {
            MCX_WAVE_LDS_SYNC();                             // column j - 1 (other lanes' rows) is written
            const double ajj = 1.0 / M[rbj + j];
            double x[NC];
#pragma unroll
            for (int q = 0; q < NC; ++q) x[q] = 0.0;
            int rb = 0;
#pragma unroll 4
            for (int jj = 0; jj < j; ++jj) {
                const double temp = M[rb + j];               // the original A(jj,j)
                const bool nz = temp != 0.0;                 // dtrmv skips a zero (the row's own sweep then leaves the zero as it is)
#pragma unroll
                for (int q = 0; q < NC; ++q) {
                    const double m = M[rbr[q] + jj];
                    const double p = temp * m, f = dfma(temp, m, x[q]);
                    x[q] = (kk[q] == jj) ? (nz ? p : temp) : ((kk[q] < jj && nz) ? f : x[q]);
                }
                rb += d - (jj + 1);
            }
            if (go) {
#pragma unroll
                for (int q = 0; q < NC; ++q) {
                    if (kk[q] < j) M[rbr[q] + j] = (-ajj) * x[q];
                    if (kk[q] == j) M[rbj + j] = ajj;
                }
            }
            rbj += d - (j + 1);
        }
        int rbi = 0;                                                        // row i
        // dlauu2: A(r,i) = A(i,i) A(r,i) + sum_{k>i} A(i,k) A(r,k) ascending k; the diagonal's sum of squares on every lane
        for (int i = 0; i < d; ++i) {
            MCX_WAVE_LDS_SYNC();
            const double aii = M[rbi + i];
            if (i < d - 1) {
                double dot = 0.0, x[NC];
#pragma unroll
                for (int q = 0; q < NC; ++q) x[q] = aii * M[rbr[q] + i];
                dot = dfma(aii, aii, dot);
#pragma unroll 4
                for (int k = i + 1; k < d; ++k) {
                    const double temp = M[rbi + k];
                    dot = dfma(temp, temp, dot);
                    const bool nz = temp != 0.0;
#pragma unroll
                    for (int q = 0; q < NC; ++q) { const double f = dfma(temp, M[rbr[q] + k], x[q]); x[q] = nz ? f : x[q]; }
                }
                if (go) {
#pragma unroll
                    for (int q = 0; q < NC; ++q) {
                        if (kk[q] < i) M[rbr[q] + i] = x[q];
                        if (kk[q] == i) M[rbi + i] = dot;
                    }
                }
            } else if (go) {
#pragma unroll
                for (int q = 0; q < NC; ++q) if (kk[q] <= i && kk[q] < d) M[rbr[q] + i] = aii * M[rbr[q] + i];
            }
            rbi += d - (i + 1);
        }
    }
    __syncthreads();
    if (flg[cc]) {                                           // (a singular factor leaves the copy of R, as potri_packed does)
        double *iCg = E.iC + (size_t)tile * P * 64 + cl0 + cc;
        const double *Mc = Mf + (size_t)cc * PS;
        for (int e = e0; e < P; e += CB * ES) {
            double v[CB];
#pragma unroll
            for (int u = 0; u < CB; ++u) { const int ee = e + u * ES; v[u] = Mc[ee < P ? ee : P - 1]; }
#pragma unroll
            for (int u = 0; u < CB; ++u) { const int ee = e + u * ES; if (ee < P) iCg[(size_t)ee * 64] = v[u]; }
        }
    }
}

// accept bytes of a launch -> the tile ballots (MCMC_savechain's repeat counts are decoded from them): one thread per
// (iteration, tile); bit l of a ballot = chain l of the tile accepted
__global__ void group_pack_kernel(EngineDev E, const uint8_t *accb, int it0, int it1)
{
    const long long n = (long long)(it1 - it0 + 1) * E.ntiles;
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n) return;
    const int i = (int)(idx / E.ntiles), tile = (int)(idx % E.ntiles), it = it0 + i;
    const uint64_t *p = (const uint64_t *)(accb + ((size_t)i * E.ntiles + tile) * 64);
    uint64_t m = 0;
#pragma unroll
    for (int w = 0; w < 8; ++w) m |= (((p[w] & 0x0101010101010101ull) * 0x0102040810204080ull) >> 56) << (8 * w);
    if (E.hist) E.wacc[(size_t)tile * E.wcap + (it % E.wcap)] = m;
    if (E.accmask) E.accmask[(size_t)(it - 1) * E.ntiles + tile] = m;
}

} // namespace mcx
