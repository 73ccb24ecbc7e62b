// rnnt_joint_impl.h -- host driver of the additive-joint path (run_gpu_joint<Tag>) as a header: one instantiation per storage type,
// each in a translation unit -- a code object -- of its own (rnnt_joint.hip: fp32 + the C entry points; rnnt_joint_bf16.hip;
// rnnt_joint_fp16.hip), so that the first additive-joint call of a process loads the kernels of ITS storage type only (the three
// together are 3.8 MB of device code; rnnt_gpu_impl.h has the measurement for the materialised path).
#pragma once
#include <atomic>

#include <type_traits>

#include "rnnt_host.h"
#include "rnnt_joint_kernels.h"
#include "rnnt_joint16_kernels.h"

namespace rnnt {

// ----------------------------------------------------------------------------- additive joint
// f (N,maxT,A) + g (N,maxU,A) -> costs, df, dg without the (N,T,U,A) tensor (rnnt_joint_kernels.h):
// the two streaming stages are replaced, lattice and coefficients are the same launches as above.
// Enqueue only: device costs, no host copy, no synchronisation.  Storage of f, g, df, dg by tag (fp32 / bf16 / fp16),
// arithmetic fp32.
// phases: 1 = forward (row maxima, Z, lattice, and with want_grad the coefficient table + W),
// 2 = backward (DF, DG, corrections from the workspace a forward call left), 3 = both.
template <typename Tag>
rnntStatus_t run_gpu_joint(const typename Tag::store* f, const typename Tag::store* g, typename Tag::store* df,
                                  typename Tag::store* dg, const int* labels,
                                  const int* label_lengths, const int* input_lengths, int A, int N,
                                  float* costs_device, const float* grad_scale, void* workspace,
                                  const rnntOptions& opt, int phases, bool want_grad, float fastemit = 0.0f) {
    using S = typename Tag::store;
    // 16-bit storage runs the three GEMMs on the bf16 matrix cores (rnnt_joint16_kernels.h).  fp32 storage does not: with the
    // operands split into THREE bf16 parts (what fp32-class results need: six MFMAs and three roundings per element) the
    // same kernels were measured at the speed of the fp32-MFMA ones (c3 shape 0.556 vs 0.545 ms, N=128,T=200,U=41,A=1024
    // 0.302 vs 0.308: profiles/r03k_joint_fp32_on_bf16_cores.log) -- the fp32 forms stay.
    constexpr bool k16 = sizeof(typename Tag::store) == 2;
    const int jbits = tune().j16;                          // which stages take the matrix-core form (bit 0 DG, 1 DF, 2 Z)
    Plan<float> p;
    if (!make_plan(p, A, N, opt, workspace, labels, label_lengths, input_lengths, costs_device, /*joint=*/true))
        return RNNT_STATUS_INVALID_VALUE;
    if (!(fastemit >= 0.0f)) return RNNT_STATUS_INVALID_VALUE;
    p.fastemit = fastemit;
    const bool do_fwd = (phases & 1) != 0, do_bwd = (phases & 2) != 0;
    if (do_bwd && (df == nullptr || dg == nullptr)) return RNNT_STATUS_INVALID_VALUE;
    // the gradient GEMMs address one sample's rows with 32-bit element offsets
    if (static_cast<long long>(p.maxT > p.maxU ? p.maxT : p.maxU) * A >= (1LL << 31) ||
        static_cast<long long>(N) * (p.maxT + p.maxU) >= (1LL << 31))
        return RNNT_STATUS_INVALID_VALUE;
    const bool training = want_grad;
    std::unique_lock<std::mutex> prof_lock;
    if (g_prof.on.load(std::memory_order_relaxed)) prof_lock = std::unique_lock<std::mutex>(g_prof_mu);
    const bool prof = prof_prepare(prof_lock.owns_lock());
    const bool ranges = ranges_prepare();
    static const char* const kStages[4] = {"warprnnt:joint_partition", "warprnnt:lattice", "warprnnt:coefficients",
                                           "warprnnt:joint_gradients"};
    auto mark = [&](int i) {
        if (prof) prof_mark(i, do_fwd, do_bwd, p.stream);
        if (ranges) ranges_mark(i, do_fwd, do_bwd, kStages);
    };

    const int maxT = p.maxT, maxU = p.maxU;
    // rows made of whole 16-byte packets (row-maximum kernel) and 4-element loads aligned (Z kernel)
    const bool vec = (A % static_cast<int>(16 / sizeof(S)) == 0) &&
                     ((reinterpret_cast<uintptr_t>(f) | reinterpret_cast<uintptr_t>(g)) & 15u) == 0;
    const int tilesT = (maxT + 31) / 32, tilesU = (maxU + 31) / 32, tiles = tilesT * tilesU;
    mark(0);
    if (do_fwd) {   // row maxima, then the partition-function GEMM with the log-prob epilogue
        const long long rows = static_cast<long long>(N) * (maxT + maxU);
        const bool per_block = static_cast<size_t>(A) * sizeof(S) >= 12288;       // long rows: a block per row
        const bool per_lanes = A <= 64;                                            // short rows: eight lanes per row
        const dim3 rgrid(static_cast<unsigned>(per_block ? rows : per_lanes ? (rows + 31) / 32 : (rows + 3) / 4));
        // vocabulary slices per tile: few tiles and a long contraction -> split it over 4 or 8 wavefronts
        const long long all_tiles = static_cast<long long>(N) * tiles;
        const int nchunk = (A + 31) / 32;
        int S = (all_tiles >= 4096 || nchunk < 16) ? 1 : ((all_tiles < 1024 && nchunk >= 32) ? 8 : 4);
        if (tune().jzs == 1 || tune().jzs == 4 || tune().jzs == 8) S = tune().jzs;
        const bool small = S == 1 && A <= kJointZSmallA && tune().jzs != 1;
        // sampled row references + guard (rnnt_joint_kernels.h): no row-maximum pass in front of the Z kernel; the exact
        // pair runs behind it only when a row tripped the guard
        const bool sampled = !small && A >= 64 && tune().jsamp != 0;
        const bool z16 = k16 && (jbits & 4) != 0 && !small && A % 8 == 0 && A >= 512 &&
                         ((reinterpret_cast<uintptr_t>(f) | reinterpret_cast<uintptr_t>(g)) & 15u) == 0;
        (void)z16;
        int* const gate = reinterpret_cast<int*>(p.rowmax + rows + 1);
        static std::atomic<int> call_counter{1};
        const int seq = call_counter.fetch_add(1, std::memory_order_relaxed);
        float* side0 = training ? p.side : nullptr;
        const unsigned nside = static_cast<unsigned>(p.side_bytes / sizeof(float));
#define RNNT_JMAX(VV, WW, SIDE, GATE)                                                                           \
    hipLaunchKernelGGL((joint_rowmax_kernel<Tag, VV, WW>), rgrid, dim3(256), 0, p.stream, f, g, input_lengths,       \
                       label_lengths, p.rowmax, maxT, maxU, A, N, SIDE, nside, GATE, seq)
#define RNNT_JMAX_ALL(SIDE, GATE)                                                                               \
        do {                                                                                                    \
            if (per_block) { if (vec) RNNT_JMAX(true, 4, SIDE, GATE); else RNNT_JMAX(false, 4, SIDE, GATE); }   \
            else if (per_lanes) RNNT_JMAX(false, 0, SIDE, GATE);                                                \
            else { if (vec) RNNT_JMAX(true, 1, SIDE, GATE); else RNNT_JMAX(false, 1, SIDE, GATE); }             \
        } while (0)
#define RNNT_JZ(SS, VV, SAMP, GATE)                                                                              \
    hipLaunchKernelGGL((joint_z_kernel<Tag, SS, VV, SAMP>), SS == 1 ? dim3(((tiles + 3) / 4 + 7) / 8 * 8, N) : dim3(xcd_shared_grid(1, tiles, N)),               \
                       dim3(SS == 1 ? 256 : SS * 64), 0, p.stream, f, g, p.rowmax, labels, input_lengths,        \
                       label_lengths, p.lp2, p.logz, maxT, maxU, p.Up, A, p.blank, tilesU, tiles, N, GATE, seq, p.poison)
#define RNNT_JZ16(SS, SAMP, GATE)                                                                                \
    hipLaunchKernelGGL((joint_z16_kernel<Tag, SS, SAMP>), SS == 1 ? dim3(((tiles + 3) / 4 + 7) / 8 * 8, N) : dim3(xcd_shared_grid(1, tiles, N)),    \
                       dim3(SS == 1 ? 256 : SS * 64), 0, p.stream, f, g, p.rowmax, labels, input_lengths,        \
                       label_lengths, p.lp2, p.logz, maxT, maxU, p.Up, A, p.blank, tilesU, tiles, N, GATE, seq, p.poison)
#define RNNT_JZ_ALL(SAMP, GATE)                                                                                  \
        do {                                                                                                    \
            if constexpr (k16) {                                                      \
                if (z16) {     /* bf16 storage on the bf16 matrix cores (rnnt_joint16_kernels.h) */              \
                    if (S == 8) RNNT_JZ16(8, SAMP, GATE); else if (S == 4) RNNT_JZ16(4, SAMP, GATE);            \
                    else RNNT_JZ16(1, SAMP, GATE);                                                              \
                    break;                                                                                      \
                }                                                                                               \
            }                                                                                                   \
            if (S == 8) { if (vec) RNNT_JZ(8, true, SAMP, GATE); else RNNT_JZ(8, false, SAMP, GATE); }          \
            else if (S == 4) { if (vec) RNNT_JZ(4, true, SAMP, GATE); else RNNT_JZ(4, false, SAMP, GATE); }     \
            else { if (vec) RNNT_JZ(1, true, SAMP, GATE); else RNNT_JZ(1, false, SAMP, GATE); }                 \
        } while (0)
        if (sampled) {
            const unsigned pgrid = (nside + 255) / 256 > 0 ? (nside + 255) / 256 : 1;
            hipLaunchKernelGGL(joint_prep_kernel<0>, dim3(pgrid), dim3(256), 0, p.stream, p.rowmax, side0, nside, gate, seq,
                               static_cast<unsigned>(rows));
            RNNT_JZ_ALL(true, gate);
            RNNT_JMAX_ALL(static_cast<float*>(nullptr), gate);           // the exact pair: returns at once unless the gate is raised
            RNNT_JZ_ALL(false, gate);
        } else {
            int* const no_gate = nullptr;                                // the row-maximum pass always runs
            RNNT_JMAX_ALL(side0, no_gate);
            p.check();
            if (small)
                hipLaunchKernelGGL((joint_z_small_kernel<Tag>), dim3(((tiles + 3) / 4 + 7) / 8 * 8, N), dim3(256),
                                   4 * joint_z_small_slice(A) * sizeof(float), p.stream, f, g, p.rowmax, labels,
                                   input_lengths, label_lengths, p.lp2, p.logz, maxT, maxU, p.Up, A, p.blank, tilesU,
                                   tiles, N, p.poison, joint_z_small_slice(A));
            else
                RNNT_JZ_ALL(false, no_gate);
        }
#undef RNNT_JMAX
#undef RNNT_JMAX_ALL
#undef RNNT_JZ
#undef RNNT_JZ_ALL
#undef RNNT_JZ16
        p.check();
    }
    mark(1);
    if (do_fwd) launch_lattice(p, training);
    mark(2);
    // small vocabularies: the df corrections ride along in the DF GEMM as one-hot operands (3x its
    // contraction) instead of one global atomic per lattice cell in the fix-up kernel
    // (fp32 storage: with 16-bit storage the column-split one-hot DF kernel needs 228 + 128 registers and runs at a third of
    // the speed -- c4 shape 0.81 vs 0.51 ms for the backward phase -- so 16-bit keeps the epilogue corrections, except:)
    // 16-bit storage, at most two 32-column groups, long label rows: there the split-contraction DF kernel without the CB
    // operand (SPLIT, BS below) fits 248 registers and the one-hot form wins as it does for fp32 (c4 shape, bf16: backward
    // 0.300 -> 0.233 ms, and the coefficient kernel writes W and CL only instead of records + two planes: 0.366 -> 0.27 ms)
    const bool onehot16 = A <= 64 && maxU >= 64 && joint_planes_onehot(maxU) == 4 && coef_is_tiled(p) && tune().jfsum &&
                          tune().jnocb && tune().jsplit;
    const bool onehot = tune().joh >= 0 ? tune().joh != 0 : (A <= 256 && (sizeof(S) == 4 || onehot16));
    // correction sums (fp32 side vectors in the workspace) for the epilogues of the gradient GEMMs
    float* sfb = p.side;
    float* sgb = sfb + static_cast<size_t>(N) * maxT;
    float* sgl = sgb + static_cast<size_t>(N) * maxU;
    int* farflag = reinterpret_cast<int*>(sgl + static_cast<size_t>(N) * maxU);
    // no record table (far cells: the mark in W, their c at the plane index): the one-hot planes, or W and CL alone behind the
    // tiled coefficient kernel that forms the sums (launch_coef: planes 4 / 5)
    const bool norec = joint_planes_onehot(maxU) == 4 && (onehot || (coef_is_tiled(p) && tune().jfsum && tune().jnocb));
    const float* cplanes = norec ? p.wmat : nullptr;
    const dim3 fixgrid((maxU + 63) / 64, (maxT + kJointFixT - 1) / kJointFixT, N);
    if (do_fwd && training) {
        const JointSums sums{sfb, sgb, sgl, farflag};
        // (the tiled coefficient kernel forms the correction sums itself; the cell-per-thread form leaves them to a pass of their own)
        if (!launch_coef(p, /*joint=*/true, onehot, tune().jfsum ? &sums : nullptr))
            hipLaunchKernelGGL(joint_sums_kernel<0>, fixgrid, dim3(256), 0, p.stream, p.rowtab, input_lengths, label_lengths, sfb,
                               sgb, sgl, farflag, maxT, maxU, N, cplanes, joint_upad(maxU));
        p.check();
    }
    mark(3);
    if (do_bwd) {
        // gradient GEMMs with the corrections in their epilogues (plain stores of every element, padding
        // included), then the far cells (rare).
        // NK adjacent columns per lane = the widest vector the vocabulary size and alignment allow.
        const int Upad = joint_upad(maxU);
        const uintptr_t all4 = reinterpret_cast<uintptr_t>(f) | reinterpret_cast<uintptr_t>(g) |
                               reinterpret_cast<uintptr_t>(df) | reinterpret_cast<uintptr_t>(dg);
        const int NKmax = (A % 4 == 0 && (all4 & (4 * sizeof(S) - 1)) == 0 && A >= 96) ? 4
                        : (A % 2 == 0 && (all4 & (2 * sizeof(S) - 1)) == 0 && A >= 48) ? 2 : 1;
        const Tune& tn = tune();
        auto pick = [&](int want) { int nk = want > 0 ? want : NKmax; while (nk > NKmax) nk >>= 1; return nk; };
        int NKf = pick(tn.jfnk);
        const int NKg = pick(tn.jgnk);
        auto groups = [&](int nk) { return (A + 32 * nk - 1) / (32 * nk); };   // 32 NK-column groups of the vocabulary
        if (tn.jfnk <= 0) {
            // DF, columns per lane: at most 64 symbols and long label rows -> ONE group of 64 columns whose contraction the
            // block's wavefronts split (below); otherwise the widest NK that still gives each of the block's four wavefronts
            // columns to work on (a block spans 128 NK columns: at A = 256, NK = 4 leaves two of them idle -- long utterances,
            // fp32: 353 us against 194 with NK = 2; 16-bit storage halves the register budget per column once more: NK = 1
            // 198 us against 245)
            if (A <= 64 && maxU >= 64) NKf = NKf < 2 ? NKf : 2;
            else while (NKf > 1 && 128 * NKf * (sizeof(S) == 2 ? 2 : 1) > A) NKf >>= 1;
        }
        // small vocabularies (one or two column groups): the four wavefronts of a block split the CONTRACTION instead of the
        // columns (joint_df_kernel, SPLIT) -- when it is long enough to be worth the reduction
        const bool nocb = onehot && joint_planes_onehot(maxU) == 4 && coef_is_tiled(p) && tn.jfsum && tn.jnocb;
        // (DF: never with four columns per lane -- that instantiation needs more than 512 registers; jsplit = 2: dev, any vocabulary)
        const bool split_f = tn.jsplit && NKf <= 2 && maxU >= 64 && (groups(NKf) <= 2 || tn.jsplit >= 2);
        // DG: also whenever the contraction over t is LONG, whatever the vocabulary -- the block count of the column-split form does
        // not grow with T (N=16, T=1500, U=301, A=1024: 320 blocks, 276 us; split 192 us), at T = 150 / 200 the split form loses 5-10 %
        const bool split_g = tn.jsplit && maxT >= 64 && (groups(NKg) <= 2 || maxT >= 512 || tn.jsplit >= 2);
#define RNNT_JDF_SPLIT(NN, OO)                                                                                   \
    hipLaunchKernelGGL((joint_df_kernel<Tag, NN, true, OO, true, false>), dim3(xcd_shared_grid((A + 32 * NN - 1) / (32 * NN), tilesT, N)), dim3(256), 0, \
                       p.stream, f, g, p.rowmax, p.wmat, grad_scale, labels, input_lengths, label_lengths, df, maxT,  \
                       maxU, Upad, A, N, p.blank, sfb)
#define RNNT_JDF_SPLIT_BS(NN)                                                                                    \
    hipLaunchKernelGGL((joint_df_kernel<Tag, NN, true, true, true, true>), dim3(xcd_shared_grid((A + 32 * NN - 1) / (32 * NN), tilesT, N)), dim3(256), 0, \
                       p.stream, f, g, p.rowmax, p.wmat, grad_scale, labels, input_lengths, label_lengths, df, maxT,  \
                       maxU, Upad, A, N, p.blank, sfb)
#define RNNT_JDF_BS(NN)                                                                                          \
    hipLaunchKernelGGL((joint_df_kernel<Tag, NN, true, true, false, true>), dim3(xcd_shared_grid((A + 128 * NN - 1) / (128 * NN), tilesT, N)), dim3(256), 0, \
                       p.stream, f, g, p.rowmax, p.wmat, grad_scale, labels, input_lengths, label_lengths, df, maxT,  \
                       maxU, Upad, A, N, p.blank, sfb)
#define RNNT_JDG_SPLIT(NN)                                                                                       \
    hipLaunchKernelGGL((joint_dg_kernel<Tag, NN, true, true>), dim3(xcd_shared_grid((A + 32 * NN - 1) / (32 * NN), tilesU, N)), dim3(256), 0, \
                       p.stream, f, g, p.rowmax, p.wmat, grad_scale, input_lengths, label_lengths, dg, maxT,     \
                       maxU, Upad, A, N, labels, p.blank, sgb, sgl)
#define RNNT_JDF(NN, PP, OO)                                                                                     \
    hipLaunchKernelGGL((joint_df_kernel<Tag, NN, PP, OO>), dim3(xcd_shared_grid((A + 128 * NN - 1) / (128 * NN), tilesT, N)), dim3(256), 0, \
                       p.stream, f, g, p.rowmax, p.wmat, grad_scale, labels, input_lengths, label_lengths, df, maxT,  \
                       maxU, Upad, A, N, p.blank, sfb)
#define RNNT_JDG(NN, PP)                                                                                         \
    hipLaunchKernelGGL((joint_dg_kernel<Tag, NN, PP>), dim3(xcd_shared_grid((A + 128 * NN - 1) / (128 * NN), tilesU, N)), dim3(256), 0, \
                       p.stream, f, g, p.rowmax, p.wmat, grad_scale, input_lengths, label_lengths, dg, maxT,     \
                       maxU, Upad, A, N, labels, p.blank, sgb, sgl)
        // 16-bit storage: the operand ping-pong doubles the AGPR count of these kernels (172 + 128 registers: one wavefront
        // per SIMD at four columns per lane); without it they keep two (DF) / three (DG) wavefronts per SIMD
        const bool pf_f = tn.jfpf != 0 && (sizeof(S) == 4 || NKf < 4), pf_g = tn.jgpf != 0 && (sizeof(S) == 4 || NKg < 4);
        bool df16 = false;
        if constexpr (k16) {
            df16 = (jbits & 2) != 0 && A % 8 == 0 && (all4 & 15u) == 0 && A >= 512;
            if (df16) {
#define RNNT_JDF16(NN, PP)                                                                                       \
    hipLaunchKernelGGL((joint_df16_kernel<Tag, NN, PP>), dim3(xcd_shared_grid((A + 128 * NN - 1) / (128 * NN), tilesT, N)), dim3(256), 0, p.stream, \
                       f, g, p.rowmax, p.wmat, grad_scale, labels, input_lengths, label_lengths, df, maxT, maxU, Upad, A, N, \
                       p.blank, sfb)
                if (tn.j16nt == 8) { if (tn.j16pf) RNNT_JDF16(8, true); else RNNT_JDF16(8, false); }
                else { if (tn.j16pf) RNNT_JDF16(4, true); else RNNT_JDF16(4, false); }
#undef RNNT_JDF16
            }
        }
        if (df16) { /* launched above */ }
        else if (split_f && onehot && nocb) { if (NKf == 2) RNNT_JDF_SPLIT_BS(2); else RNNT_JDF_SPLIT_BS(1); }
        else if (onehot && nocb) { if (NKf == 4) RNNT_JDF_BS(4); else if (NKf == 2) RNNT_JDF_BS(2); else RNNT_JDF_BS(1); }
        else if (split_f && onehot) { if (NKf == 2) RNNT_JDF_SPLIT(2, true); else RNNT_JDF_SPLIT(1, true); }
        else if (split_f) { if (NKf == 2) RNNT_JDF_SPLIT(2, false); else RNNT_JDF_SPLIT(1, false); }
        else if (onehot)  { if (NKf == 4) RNNT_JDF(4, true, true); else if (NKf == 2) RNNT_JDF(2, true, true); else RNNT_JDF(1, true, true); }
        else if (pf_f) { if (NKf == 4) RNNT_JDF(4, true, false); else if (NKf == 2) RNNT_JDF(2, true, false); else RNNT_JDF(1, true, false); }
        else              { if (NKf == 4) RNNT_JDF(4, false, false); else if (NKf == 2) RNNT_JDF(2, false, false); else RNNT_JDF(1, false, false); }
        // bf16 storage, rows of whole 16-byte packets: the bf16 matrix-core forms (rnnt_joint16_kernels.h)
        bool dg16 = false;
        if constexpr (k16) {
            dg16 = (jbits & 1) != 0 && A % 8 == 0 && (all4 & 15u) == 0 && A >= 512;
            if (dg16) {
#define RNNT_JDG16(NN, PP)                                                                                       \
    hipLaunchKernelGGL((joint_dg16_kernel<Tag, NN, PP>), dim3(xcd_shared_grid((A + 128 * NN - 1) / (128 * NN), tilesU, N)), dim3(256), 0, p.stream, \
                       f, g, p.rowmax, p.wmat, grad_scale, input_lengths, label_lengths, dg, maxT, maxU, Upad, A, N,  \
                       labels, p.blank, sgb, sgl)
                if (tn.j16nt == 8) { if (tn.j16pf) RNNT_JDG16(8, true); else RNNT_JDG16(8, false); }
                else { if (tn.j16pf) RNNT_JDG16(4, true); else RNNT_JDG16(4, false); }
#undef RNNT_JDG16
            }
        }
        if (dg16) { /* launched above */ }
        else if (split_g) { if (NKg == 4) RNNT_JDG_SPLIT(4); else if (NKg == 2) RNNT_JDG_SPLIT(2); else RNNT_JDG_SPLIT(1); }
        else if (pf_g) { if (NKg == 4) RNNT_JDG(4, true); else if (NKg == 2) RNNT_JDG(2, true); else RNNT_JDG(1, true); }
        else         { if (NKg == 4) RNNT_JDG(4, false); else if (NKg == 2) RNNT_JDG(2, false); else RNNT_JDG(1, false); }
#undef RNNT_JDF
#undef RNNT_JDG
#undef RNNT_JDF_SPLIT
#undef RNNT_JDF_SPLIT_BS
#undef RNNT_JDF_BS
#undef RNNT_JDG_SPLIT
        p.check();
        if constexpr (sizeof(S) == 4) {
            hipLaunchKernelGGL((joint_far_kernel<Tag>), fixgrid, dim3(256), 0, p.stream, f, g, p.rowmax, p.rowtab, grad_scale,
                               input_lengths, label_lengths, farflag, df, dg, maxT, maxU, A, N, cplanes, Upad);
        } else {
            // 16-bit gradients: the far cells of a row segment are summed in fp32 before the stored element is touched (joint_far16_kernel)
            const int longer = maxT > maxU ? maxT : maxU;
            const dim3 fargrid((longer + 63) / 64, (maxT + kJointFixT - 1) / kJointFixT + (maxU + kJointFixT - 1) / kJointFixT, N);
            hipLaunchKernelGGL((joint_far16_kernel<Tag>), fargrid, dim3(256), 0, p.stream, f, g, p.rowmax, p.rowtab, grad_scale,
                               input_lengths, label_lengths, farflag, df, dg, maxT, maxU, A, N, cplanes, Upad);
        }
        p.check();
    }
    mark(4);
    if (prof) g_prof.pending = true;
    return p.failed ? RNNT_STATUS_EXECUTION_FAILED : RNNT_STATUS_SUCCESS;
}

}  // namespace rnnt

namespace rnnt {
#ifndef RNNT_JOINT_INSTANTIATE_F32
extern template rnntStatus_t run_gpu_joint<F32>(const float*, const float*, float*, float*, const int*, const int*, const int*, int, int, float*, const float*, void*, const rnntOptions&, int, bool, float);
#endif
#ifndef RNNT_JOINT_INSTANTIATE_BF16
extern template rnntStatus_t run_gpu_joint<BF16>(const uint16_t*, const uint16_t*, uint16_t*, uint16_t*, const int*, const int*, const int*, int, int, float*, const float*, void*, const rnntOptions&, int, bool, float);
#endif
#ifndef RNNT_JOINT_INSTANTIATE_FP16
extern template rnntStatus_t run_gpu_joint<F16>(const uint16_t*, const uint16_t*, uint16_t*, uint16_t*, const int*, const int*, const int*, int, int, float*, const float*, void*, const rnntOptions&, int, bool, float);
#endif
}  // namespace rnnt
