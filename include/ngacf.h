/*
 * ngacf.h -- C ABI of the MI355X (gfx950) graph-attention propagation library.
 *
 * The reference (cleverer123/NGACF) has no FFI: its hot path is a chain of ATen
 * calls inside nn.Module.forward.  Each entry point below replaces one such chain;
 * the citation names the reference lines it stands in for (paths relative to the
 * reference root).  INTEGRATION.md shows the ctypes binding a maintainer of the
 * reference would add.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer unless stated; the library owns nothing;
 *   - `stream` is a hipStream_t passed as void*; every call is asynchronous on it;
 *   - return value: 0 on success, a negative NGACF_E_* code for a rejected
 *     argument (nothing was launched), or a positive hipError_t from the launch;
 *   - no global mutable state: safe from several host threads on different devices;
 *   - dropout seeds: effective seed = `seed` + (*seed_dev if seed_dev != NULL); the device-side addend lets a
 *     captured hipGraph draw fresh masks on every replay (the host advances it with a captured add).
 *
 * Node numbering: users are 0..U-1, items are U..U+I-1 (N = U+I), exactly the
 * `features = cat([userEmbd, itemEmbd])` order of graphattention/SPUIGACF.py:38-39.
 *
 * Graph plan (built once per adjacency by the host, see ngacf_amd/plan.py):
 *   col   int32[2E]      source node (global id) of every directed edge; the 2E entries
 *                        are the CSR of the user rows followed by the CSC of the item rows
 *   rowptr int32[N+1]    edge range of every dst row in `col`
 *   srow  int32[nshort][4] the short dst rows (degree <= plan.short_max), one record {row, first edge, degree, 0} each (the
 *                        row's slice of rowptr, copied: a wave learns row and edge range with one 16-byte load): traversed one
 *                        lane group per row.  First the user rows (sorted by degree, descending, padded with records of row -1
 *                        and degree 0 to a multiple of 16 = nshort_user entries), then the item rows (sorted likewise)
 *   hrow  int32[nhub]    the hub rows (plan.seg_len < degree <= plan.hub_cap): ONE WORKGROUP per row, its four waves
 *                        stride over the row's edges and fold through LDS; the first nhub_user entries are user rows
 *   nseg_user            number of leading entries of `segs` that are user rows; a launch works through
 *                        [user hubs][user segs][user short rows][item hubs][item segs][item short rows], so each
 *                        half re-reads only ONE of the two node tables and that table stays L2 resident
 *   segs  int32[nseg][4] work items of the long rows {dst row, first edge, one-past-last edge, part}:
 *                        part < 0  -> the segment is the whole row (degree <= seg_len), finished in place;
 *                        part >= 0 -> slice of a row beyond hub_cap, partial sums go to slot `part`
 *   split int32[nsplit][3] {dst row, first part slot, number of slots} for the rows cut into slices
 */
#ifndef NGACF_H
#define NGACF_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define NGACF_F32  0
#define NGACF_BF16 1

/* flags of ngacf_prop_* */
#define NGACF_ELU        1   /* apply ELU to h + agg (SPUIGACF.py:395-397 and the outer F.elu :214) */
#define NGACF_RESIDUAL   2   /* out = h_dst + agg (SPUIGACF.py:383,390); clear for SPGA.py:405-409 */
#define NGACF_SKIP_NODE  8   /* ngacf_prop_bwd: step 1 was already run by ngacf_prop_bwd_node */
#define NGACF_GOUT_F32  16  /* ngacf_prop_bwd: gout is float[N][d] whatever the storage dtype (the scored batch's gradient rows, ngacf_score_bwd) */
#define NGACF_SCORE_ROW  4   /* ngacf_prop_fwd: recompute the source score a_src . h_src from the gathered row
                                instead of reading S_src (saves 4*heads bytes per edge); ngacf_prop_bwd ignores it: it gathers
                                the source's score with t as one (S, t) pair */
/* MEASUREMENT ONLY (tools/gather_roof.py; d == 64, eval / dense launches without dropout, row flags or fused tails): the traversal
   kernel stripped to its memory operations -- same work order, lane layout, index loads, row (and (S, t) pair) gathers, loop
   structure and stores; per gathered element one add (ROOF_ADD) or per gathered dword one xor (ROOF_XOR); no score, exp, mask,
   division, ELU.  The outputs are meaningless; the launch's DURATION is the measured gather roof bench.py grades the kernels
   against (roofline.measured_gather_roof_GBps). */
#define NGACF_ROOF_ADD  32
#define NGACF_ROOF_XOR  64

#define NGACF_TBWD_BLOCKS 1024 /* workgroups of the dW reduction in ngacf_transform_bwd */

#define NGACF_E_ARG     (-1)  /* null pointer / negative size */
#define NGACF_E_SHAPE   (-2)  /* unsupported (d, heads, dtype) combination */

const char* ngacf_version(void);

/* Row transform + per-node attention scores.
 * Replaces torch.mm(features[uidx], W_u), torch.mm(features[iidx], W_i) (SPUIGACF.py:356-357)
 * for all heads of a stage at once, the feature dropout in front of the stage (:208,:213), and
 * the node halves of a.mm(edge_h) (:359-361):  S[n,h] = sum_{c in head h} H[n,c] * avec[type(n)][c].
 *   X     [N,din]   x_dtype          W    float[2][din][dout]  (0 = W_u heads side by side, 1 = W_i)
 *   avec  float[2][dout]  (0 = the a[:, :k] halves of the heads, 1 = the a[:, k:] halves)
 *   H     [N,dout]  h_dtype          S    float[N][heads]
 * feat_drop_p > 0 multiplies X by the counter-hash dropout mask keyed by (seed, n, c). */
int ngacf_transform_fwd(const void* X, int x_dtype, const float* W, const float* avec,
                        void* H, int h_dtype, float* S,
                        int U, int I, int din, int dout, int heads,
                        float feat_drop_p, uint64_t seed, const uint64_t* seed_dev, void* stream);

/* ngacf_transform_fwd on the matrix-core path (h_dtype must be NGACF_BF16) whose launch ALSO packs the weight panels of up to
 * NGACF_MAX_PACK later stages for ngacf_prop_fwd_next (side workgroups of the same launch: the first launch of a forward
 * refreshes the fragments from the live fp32 weights, so a captured hipGraph never replays a stale copy after an optimizer
 * step).  Host arrays of n_pack entries: pack_W[k] float[2][din_k][dout_k] (device), pack_panel[k] device buffer of
 * 2 * din_k * dout_k * 2 bytes, din_k == 64, dout_k in {64, 128}.  Fragment layout (16 bytes = 8 bf16 each):
 *   index ((y * (din/16) + s) * (dout/32) + t) * 64 + l  holds  W[y][16 s + 8 (l >> 5) + j][32 t + (l & 31)], j = 0..7
 *   (the A fragments of v_mfma_f32_32x32x16_bf16 with the output column on the A operand: csrc/mfma_rows.hpp).
 * pack_kind (host array, or NULL = all 0): 0 = the forward fragments above; 1 = the fragments of the TRANSPOSED product
 * dX = dH @ W^T for ngacf_prop_bwd_dx (din_k == dout_k == 64):
 *   index ((y * (dout/16) + s) * (din/32) + t) * 64 + l  holds  W[y][32 t + (l & 31)][16 s + 8 (l >> 5) + j], j = 0..7.
 * U == I == 0: only the panels are packed (X .. S are not read; the node-partitioned path, whose transforms run over row windows). */
#define NGACF_MAX_PACK 8
int ngacf_transform_fwd_pack(const void* X, int x_dtype, const float* W, const float* avec,
                             void* H, int h_dtype, float* S,
                             int U, int I, int din, int dout, int heads,
                             float feat_drop_p, uint64_t seed, const uint64_t* seed_dev,
                             int n_pack, const float* const* pack_W, void* const* pack_panel,
                             const int* pack_din, const int* pack_dout, const int* pack_kind, void* stream);

/* The same over row windows (node-partitioned runs, ngacf_amd/dist.py): users [u_lo, u_lo+u_n) and items
 * [i_lo, i_lo+i_n) (global row = U + i).  H and S are indexed by global row; with x_local != 0, X holds only
 * the window rows (the u_n user rows first, then the i_n item rows). */
int ngacf_transform_fwd_rows(const void* X, int x_dtype, const float* W, const float* avec,
                             void* H, int h_dtype, float* S,
                             int U, int I, int din, int dout, int heads,
                             int u_lo, int u_n, int i_lo, int i_n, int x_local,
                             float feat_drop_p, uint64_t seed, const uint64_t* seed_dev, void* stream);

/* Attention propagation, forward, both directions in one launch.
 * Replaces SPUIGACF.py:359-393 (edge gather, exp(-leakyrelu), e_rowsum/e_colsum, edge dropout,
 * the two numerator sparse.mm, div, NaN->0, residual) plus the ELU of :397/:214.
 *   H      [N,d] dtype     S_dst,S_src float[N][heads] (may alias; S_src unused with NGACF_SCORE_ROW)
 *   avec   float[2][d]     (as in ngacf_transform_fwd; used only with NGACF_SCORE_ROW)
 *   out    [N,d] dtype     agg [2][N][d] dtype / R float[2][N][heads], saved for the backward (both NULL, or both given =
 *                          a train-mode forward): plane 0 = the normalised aggregate (in the storage type, like every other
 *                          table of the stage) and the denominators; plane 1 = the same sums restricted to the edges whose
 *                          score s = S[dst] + S[src] is not positive (aggn = sum m e H[src] / R, Rn = sum e over them), which
 *                          make the "own gradient" half of the backward's dS row-local (ngacf_prop_bwd)
 *   part_acc float[nparts][d], part_r float[nparts][heads]: workspace for hub rows; with agg / R given a slot holds both
 *                          planes: float[nparts][2][d], float[nparts][2][heads]
 *   part_split int32[nparts] (hub row of every slot), tickets int32[nsplit] (all zero on entry, left all zero):
 *       with both given, the last-arriving slice of a hub row finishes it inside the launch (write-through stores,
 *       an agent-scope ticket, agent-scope atomic loads; no fence, no waiting); with NULL a small second kernel does (same
 *       sums, same order, same bits).  Overlapping launches (different streams) need their own tickets.
 * drop_p > 0 applies the edge dropout of :375 (numerators only), mask keyed by
 * (seed, head, user, item) so both directions see the same mask.
 * needed (uint32[ceil(N/32)], bit `row` of the table, or NULL) + rows (int32[nrows]) + short_max: compute only the flagged rows and leave the
 *   others' out / agg / R untouched.  SPUIGACF.forward reads B user rows and B item rows of the LAST stage's output
 *   (:49-52); its other ~94 % rows are never used, so the last stage of a scored batch runs with the batch's flags and row
 *   list (both from ngacf_mark_rows; duplicates and -1 entries allowed in the list).  Rows of degree <= short_max (the
 *   plan's short_max) are taken from the list, a WAVE each (their few edges are gathered in one or two rounds); longer rows
 *   from the plan's segments, by flag.  The same terms as the full launch, added in another order for the short rows
 *   (deterministic; equal to rounding).  srow / nshort are not read.
 *   Needs NGACF_SCORE_ROW.  (In-kernel hub finish: all slices of a flagged hub row run, none of an unflagged one, so the
 *   arrival counters work unchanged.) */
int ngacf_prop_fwd(const int32_t* segs, int nseg, const int32_t* split, int nsplit,
                   const int32_t* srow, int nshort, int nseg_user, int nshort_user,
                   const int32_t* hrow, int nhub, int nhub_user, const int32_t* rowptr,
                   const int32_t* col, const void* H, int dtype,
                   const float* S_dst, const float* S_src, const float* avec,
                   void* out, void* agg, float* R, float* part_acc, float* part_r,
                   const int32_t* part_split, int32_t* tickets,
                   const uint32_t* needed, const int32_t* rows, int nrows, int short_max,
                   int U, int N, int d, int heads, float alpha,
                   float drop_p, uint64_t seed, const uint64_t* seed_dev, int flags, void* stream);

/* The NEXT stage's row transform fused into the traversal: SpUIGAT.forward chains the stages (SPUIGACF.py:207-215: the heads'
 * outputs are concatenated, dropped out and fed to the next layer's torch.mm(features, W) :356-357), so a finished output
 * row of stage s is the input row of stage s+1's transform.  ngacf_prop_fwd_next = ngacf_prop_fwd + that transform in the
 * epilogue: every wave turns the rows it has just finished (still in registers) into
 *     H'[n, :] = dropout'(out[n, :]) @ W'[type(n)],   S'[n, h] = sum_{c in head h} H'[n, c] * avec'[type(n)][c]
 * on the matrix cores (v_mfma_f32_16x16x32_bf16; operands = the bf16 values that `out` stores, fp32 accumulate, H' rounded
 * to bf16 and S' formed from the rounded values: what ngacf_transform_fwd computes from the stored table).  One dependent
 * launch and one write + read of the [N][d] table less per stage.
 *   next->panel   bf16 storage: the next stage's W as bf16 matrix-core fragments, written by ngacf_transform_fwd_pack
 *                 (2 * din * dout * 2 bytes); fp32 storage: the next stage's W itself, float[2][64][dout'] (exact fp32 matrix
 *                 cores, v_mfma_f32_16x16x4_f32: the stand-alone fp32 transform's fmaf chains, the same bits)
 *   next->avec    float[2][dout']        next->H  [N][dout'] in the stage's storage type        next->S  float[N][heads']
 *   next->dout in {64, 128}, next->heads in {1, dout' / 8}; next->feat_drop_p / next->seed: the next stage's feature dropout
 *   (mask keyed by (seed', n, c) exactly as in ngacf_transform_fwd; the device addend seed_dev is shared with this stage).
 * Restrictions: d == 64, heads == 8, NGACF_SCORE_ROW, no `needed`, no workgroup-per-row hubs (nhub == 0), and
 * hub rows finished in-kernel (part_split + tickets given) or absent.  `out` may be NULL (an evaluation forward needs only
 * H' and S'); agg / R as in ngacf_prop_fwd.  next == NULL: exactly ngacf_prop_fwd. */
typedef struct ngacf_next_stage {
    const void* panel;
    const float* avec;
    void* H;
    float* S;
    int32_t dout, heads;
    float feat_drop_p;
    uint64_t seed;
} ngacf_next_stage;
int ngacf_prop_fwd_next(const int32_t* segs, int nseg, const int32_t* split, int nsplit,
                        const int32_t* srow, int nshort, int nseg_user, int nshort_user,
                        const int32_t* hrow, int nhub, int nhub_user, const int32_t* rowptr,
                        const int32_t* col, const void* H, int dtype,
                        const float* S_dst, const float* S_src, const float* avec,
                        void* out, void* agg, float* R, float* part_acc, float* part_r,
                        const int32_t* part_split, int32_t* tickets,
                        const uint32_t* needed, const int32_t* rows, int nrows, int short_max,
                        int U, int N, int d, int heads, float alpha,
                        float drop_p, uint64_t seed, const uint64_t* seed_dev, int flags,
                        const ngacf_next_stage* next, void* stream);

/* Backward of ngacf_prop_fwd (autograd of SPUIGACF.py:359-397; formulas SURVEY.md §8(a) A4b, regrouped).
 *   With e = exp(-lrelu(s)), s = S[n] + S[c], m the edge's dropout multiplier, the gradient of a score is
 *       ds(n,c) = -lrelu'(s) e (m (Gn[n].H[c] + Gn[c].H[n])_h - t[n] - t[c])
 *   and dS[n,h] = sum_c ds(n,c) is computed as P1[n,h] + the edge sums below, where the part that carries the row's own gradient,
 *       P1[n,h] = sum_c -lrelu'(s) e (m Gn[n].H[c] - t[n]) = (1 - alpha) (R (Gn[n].aggn[n])_h - Rn t[n]),
 *   is row-local thanks to the forward's second planes (lrelu' = 1 where s > 0, alpha elsewhere; the s > 0 terms cancel).
 * Step 1 (per node):  g = gout * ELU'(out);  Gn = g / R;  t[n,h] = (Gn . agg)_h;  ST[n,h] = (S[n,h], t[n,h]);  dS[n,h] = P1[n,h]
 * Step 2 (per edge, pull):  dH[n] = g[n] + sum_c m e Gn[c] + dS[n,h] * avec[type(n)]
 *                           dS[n,h] += sum_c -e * lrelu'(s) * (m (Gn[c].H[n])_h - t[c])
 *   so an edge gathers ONE row (Gn[c]) and 8 bytes per head (ST[c,h]) -- not H[c] as well.
 *   gout [N,d] dtype (the gradient arrives in the storage type of `out`)   out [N,d] dtype (forward output)
 *   agg [2][N][d], R float[2][N][heads]: as saved by the forward;  S_src float[N][heads]: the scores (step 1 copies them into ST)
 *   G float[N][d], Gn [N][d] dtype, ST float[N][heads][2] (pairs (S, t)): workspace
 *   dH [N][d] dtype (H's storage type), dS float[N][heads]: results (dS is written by step 1 and finished by step 2: with
 *   NGACF_SKIP_NODE it is an INPUT too, holding P1)
 *   G may be NULL when gout and out are given as global-row tables: step 2 then recomputes g from gout and out, bit for bit,
 *   and the 4-byte-per-element table is neither written nor read back (also with NGACF_SKIP_NODE, when Gn / ST / P1 come from elsewhere)
 *   part_acc float[nparts][d], part_r float[nparts][heads]: workspace for hub rows; part_split / tickets: as in ngacf_prop_fwd
 *   (both given: the last-arriving slice of a hub row finishes it inside the launch; NULL: a second kernel does)
 *   needed / rows / nrows (NULL / NULL / 0, or the forward's): the forward ran with these flags, i.e. gout is zero outside the
 *   flagged rows (and out / agg / R are not valid there).  Then Gn, t and P1 vanish outside and only edges whose SOURCE is
 *   flagged contribute anything -- to any row, flagged or not.  Step 1 runs over the list only; in step 2 every row tests one flag
 *   bit per edge, queues the columns that are in the batch (LDS) and gathers only those.  Equal to the unflagged call on a gout
 *   that is zero outside, bit for bit (terms that are exactly zero are left out).  With NGACF_GOUT_F32 gout is float[N][d]
 *   (only the batch's rows are read).  S_dst == S_src in every caller (one score table); avec is the node's own attention vector. */
int ngacf_prop_bwd(const int32_t* segs, int nseg, const int32_t* split, int nsplit,
                   const int32_t* srow, int nshort, int nseg_user, int nshort_user,
                   const int32_t* hrow, int nhub, int nhub_user, const int32_t* rowptr,
                   const int32_t* col, const void* H, int dtype,
                   const float* S_dst, const float* S_src, const float* avec,
                   const void* out, const void* agg, const float* R, const void* gout,
                   float* G, void* Gn, float* ST, void* dH, float* dS,
                   float* part_acc, float* part_r, const int32_t* part_split, int32_t* tickets,
                   const uint32_t* needed, const int32_t* rows, int nrows,
                   int U, int N, int d, int heads, float alpha,
                   float drop_p, uint64_t seed, const uint64_t* seed_dev, int flags, void* stream);

/* ngacf_prop_bwd whose launch ALSO runs the stage's transform backward for the rows it finishes -- the mirror image of
 * ngacf_prop_fwd_next (autograd of SPUIGACF.py:356-357 chained through :207-215; the reference's autograd runs these as separate
 * ATen kernels):
 *   dX[n, :] = mask * (dH[n, :] @ W[type(n)]^T)      what ngacf_transform_bwd's dX half computes from the stored dH, bit for bit
 *   panel   bf16 A fragments of W[type]^T: ngacf_transform_fwd_pack with pack_kind 1 (2 * din * d * 2 bytes)
 *   dX      [N][din], NGACF_BF16 or NGACF_F32 (X's type; NGACF_F32 e.g. for the first stage: the gradient of the fp32 embedding table)
 *   feat_drop_p, seed: the stage's feature dropout (the mask of ngacf_transform_fwd)
 * dH and dS are written as by ngacf_prop_bwd (the weights' gradient still needs them: ngacf_transform_bwd with dX == NULL).
 * Restrictions: dtype == NGACF_BF16, d == 64, din == 64, G == NULL, nhub == 0, hub rows finished in-kernel
 * (part_split + tickets) or absent.  dx == NULL: exactly ngacf_prop_bwd.
 * (Round 4's form also ran step 1 of the PREVIOUS stage's backward on the dX row in the same tail: worth nothing inside the
 * two-chain train step, deleted in round 5.) */
typedef struct ngacf_bwd_dx {
    const void* panel;
    void* dX;
    int32_t din, dx_dtype;
    float feat_drop_p;
    uint64_t seed;
} ngacf_bwd_dx;
int ngacf_prop_bwd_dx(const int32_t* segs, int nseg, const int32_t* split, int nsplit,
                      const int32_t* srow, int nshort, int nseg_user, int nshort_user,
                      const int32_t* hrow, int nhub, int nhub_user, const int32_t* rowptr,
                      const int32_t* col, const void* H, int dtype,
                      const float* S_dst, const float* S_src, const float* avec,
                      const void* out, const void* agg, const float* R, const void* gout,
                      float* G, void* Gn, float* ST, void* dH, float* dS,
                      float* part_acc, float* part_r, const int32_t* part_split, int32_t* tickets,
                      const uint32_t* needed, const int32_t* rows, int nrows,
                      int U, int N, int d, int heads, float alpha,
                      float drop_p, uint64_t seed, const uint64_t* seed_dev, int flags,
                      const ngacf_bwd_dx* dx, void* stream);

/* Sets bit u[b] and bit U + i[b] of needed (uint32[ceil((U+I)/32)], ALLOCATED up to the next multiple of 16 words: the clear
 * writes whole 64-byte units) for b < B (after zeroing the table when clear != 0): the rows a scored batch reads
 * from the last stage (SPUIGACF.py:49-52), for the `needed` argument of ngacf_prop_fwd / ngacf_prop_bwd; and the same rows as
 * a list: rows[b] = u[b], rows[B + b] = U + i[b] (2B entries; -1 for an id out of range).  u == NULL: items only, rows[b] = U +
 * i[b] (B entries) -- the negatives of a (user, positive, negative) batch, appended to the list of the first call.
 * gzero (float[U+I][d] or NULL): the same rows of that table are set to zero -- the rows ngacf_score_bwd will add the batch's
 * gradient to (ngacf_prop_bwd then reads them with NGACF_GOUT_F32; no other row of the table is ever read). */
int ngacf_mark_rows(const int64_t* u, const int64_t* i, uint32_t* needed, int32_t* rows, float* gzero, int B, int U, int I, int d,
                    int clear, void* stream);

/* Backward of ngacf_transform_fwd (autograd of SPUIGACF.py:356-357 and the a.mm of :361).
 *   dX   [N][din] x_dtype       = mask * (dH @ W[type]^T)   (stored in X's type: it is the gradient of X)
 *   dW   float[2][din][dout]    = (mask * X)[type rows]^T @ dH[type rows]
 *   davec float[2][dout]        = sum_n dS[n,head(c)] * H[n,c]
 *   work  float[NGACF_TBWD_BLOCKS][din*dout + dout]: per-workgroup partial sums, reduced in
 *         workgroup order (no atomics: results are bitwise reproducible)
 * dH [N][dout] is in H's storage type h_dtype, as ngacf_prop_bwd writes it (a bf16 stage's dH was rounded to bf16 on its way
 * into the matrix cores anyway: storing it that way halves two 18 MB streams); it already contains the dS*avec term.
 * Either half may be skipped: dX == NULL computes only dW/davec, dW == NULL (davec, work ignored) only dX, so a caller
 * can put the two independent halves on different streams. */
int ngacf_transform_bwd(const void* X, int x_dtype, const float* W, const void* H, int h_dtype,
                        const void* dH, const float* dS,
                        void* dX, float* dW, float* davec, float* work,
                        int U, int I, int din, int dout, int heads,
                        float feat_drop_p, uint64_t seed, const uint64_t* seed_dev, void* stream);

/* The weight halves (dW, davec) of ngacf_transform_bwd for up to NGACF_DW_BATCH_MAX stages in ONE launch plus one reduction --
 * 64 -> 64 stages, bf16 storage (bf16 MFMA) or fp32 storage (exact, on v_mfma_f32_16x16x4_f32).  A train step needs these sums only
 * at the optimizer, so the per-stage launches between the edge backwards (each latency-bound at 18 MB) collapse into one behind
 * the chain's last edge backward.  jobs: HOST array (copied into the kernel's arguments).
 *   work float[njobs][work_blocks][64*64 + 64]: per-workgroup partial sums, reduced in workgroup order (bitwise reproducible;
 *   the sums' order differs from ngacf_transform_bwd's when work_blocks differs from its 384 workgroups). */
#define NGACF_DW_BATCH_MAX 6
typedef struct ngacf_dw_job {
    const void* X;        /* [N][64] x_dtype: the stage's input */
    const void* H;        /* [N][64] h_dtype */
    const void* dH;       /* [N][64] h_dtype */
    const float* dS;      /* [N][heads] */
    float* dW;            /* [2][64][64] */
    float* davec;         /* [2][64] */
    int32_t x_dtype, h_dtype, din, dout, heads;    /* h_dtype: the type of H and dH (one per launch); fp32 storage: x_dtype is fp32 too */
    float feat_drop_p;
    uint64_t seed;
} ngacf_dw_job;
int ngacf_transform_dw_batch(const ngacf_dw_job* jobs, int njobs, float* work, int work_blocks,
                             int U, int I, const uint64_t* seed_dev, void* stream);

/* fp32 storage, a 64 -> 64 stage whose input is the previous stage's output: the dX half of ngacf_transform_bwd,
 *   dX[n, :] = mask * (dH[n, :] @ W[type(n)]^T),
 * with step 1 of the PREVIOUS stage's backward (ngacf_prop_bwd_node: Gn, the (S, t) pairs, P1) on every dX row in the same
 * launch -- the fp32 counterpart of ngacf_prop_bwd_dx's prev_* (autograd of SPUIGACF.py:356-357 chained through :207-215, :382-397);
 * bit for bit what ngacf_transform_bwd + ngacf_prop_bwd_node compute through the stored dX.
 *   prev->out [N][64], agg [2][N][64], R float[2][N][8], S float[N][8]: as the previous stage's forward left them (fp32, 8-column heads)
 *   prev->Gn [N][64], ST float[N][8][2], dS float[N][8]: results;  alpha: that stage's leaky-relu slope;  flags: its NGACF_ELU */
typedef struct ngacf_dx_prev {
    const void* out;
    const void* agg;
    const float* R;
    const float* S;
    void* Gn;
    float* ST;
    float* dS;
    float alpha;
    int32_t flags;
} ngacf_dx_prev;
int ngacf_transform_dx_node(const float* dH, const float* W, float* dX, int U, int I,
                            float feat_drop_p, uint64_t seed, const uint64_t* seed_dev,
                            const ngacf_dx_prev* prev, void* stream);

/* Row-window form (see ngacf_transform_fwd_rows): X and dX local when x_local != 0; dW and davec are the
 * sums over the window rows only (all-reduce them across ranks). */
int ngacf_transform_bwd_rows(const void* X, int x_dtype, const float* W, const void* H, int h_dtype,
                             const void* dH, const float* dS,
                             void* dX, float* dW, float* davec, float* work,
                             int U, int I, int din, int dout, int heads,
                             int u_lo, int u_n, int i_lo, int i_n, int x_local,
                             float feat_drop_p, uint64_t seed, const uint64_t* seed_dev, void* stream);

/* Step 1 of ngacf_prop_bwd alone, over row windows; gout holds only the window rows when gout_local != 0.
 * Lets a node-partitioned run all-gather Gn / ST between the two steps (then ngacf_prop_bwd with NGACF_SKIP_NODE; P1 in dS is
 * needed for the rank's own rows only).  All tables are indexed by global row: agg [2][N][d], R float[2][N][heads]. */
int ngacf_prop_bwd_node(const void* out, int dtype, const void* agg, const float* R, const void* gout,
                        const float* S_src, float* G, void* Gn, float* ST, float* dS,
                        int U, int N, int d, int heads, float alpha, int flags,
                        int u_lo, int u_n, int i_lo, int i_n, int gout_local, void* stream);

/* Pair scoring: score[b] = sum_c F[u[b],c] * F[U + i[b],c]   (SPUIGACF.py:49-52).
 * Index safety: the reference's advanced indexing raises on an out-of-range id (:49-50).  Here a pair with
 * u[b] outside [0,U) or i[b] outside [0,I) is never turned into an address: its score is NaN, the backward skips it,
 * and each such pair adds 1 to *bad_index (int32 device counter, may be NULL) for the host to raise on. */
int ngacf_score_fwd(const void* F, int dtype, const int64_t* u, const int64_t* i,
                    float* score, int32_t* bad_index, int B, int U, int I, int d, void* stream);
/* gF[u[b]] += g[b] * F[U+i[b]],  gF[U+i[b]] += g[b] * F[u[b]]   (gF float[N][d]).  B <= 4096: NO atomics -- the first pair that
 * names a row adds the contributions of all the pairs naming it, in ascending pair order, and does the row's one read-modify-write
 * (bit-reproducible; the rows named by a batch must not be written by anything else in flight).  Larger batches: float atomics. */
int ngacf_score_bwd(const void* F, int dtype, const int64_t* u, const int64_t* i,
                    const float* gscore, float* gF, int32_t* bad_index, int B, int U, int I, int d, void* stream);

/* Node-partitioned scoring (replaces the reference's replicas, parallel.py:94-196, for SPUIGACF.py:49-52 on a table whose rows
 * are spread over the ranks): the 2B rows a batch reads, contributed by their owners.
 *   ngacf_shard_rows:   rows float[2B][d]: row b = the fp32 copy of user u[b]'s (b < B) / item i[b - B]'s row of this rank's local
 *                       tables Tu [cu][d] / Ti [ci][d] when this rank owns it (internal id in [u_lo, u_lo + cu) / [i_lo, i_lo + ci)),
 *                       zeros otherwise -- the all-reduce of `rows` over the ranks then holds every row once;
 *                       pos int32[2B] = the local row (users first, an item's behind the cu user rows) or -1;
 *                       perm_u int64[U] / perm_i int64[I]: data-set id -> internal id, or NULL (identity).
 *   ngacf_rows_scatter: du [cu][d] / di [ci][d] (dtype; zero-filled by the caller): row pos[b] = sum of g[b'] (float[n][d]) over
 *                       the b' with pos[b'] == pos[b], ascending b' -- no atomics, bit-reproducible; n <= 16384. */
int ngacf_shard_rows(const void* Tu, const void* Ti, int dtype, const int64_t* u, const int64_t* i, const int64_t* perm_u,
                     const int64_t* perm_i, float* rows, int32_t* pos, int32_t* bad_index, int B, int U, int I,
                     int u_lo, int cu, int i_lo, int ci, int d, void* stream);
int ngacf_rows_scatter(const int32_t* pos, const float* g, void* du, void* di, int dtype, int n, int cu, int d, void* stream);

/* The same scatter for a SCORED batch, with the scores' backward inside it: rows = the 2B exchanged rows of ngacf_shard_rows (after the
 * all-reduce: rows[b] the user row of pair b, rows[B + b] its item row; score[b] = rows[b] . rows[B + b], SPUIGACF.py:49-52), gscore[b] =
 * d loss / d score[b]; entry k's gradient row gscore[k mod B] * rows[(k + B) mod 2B] is formed on the fly.  The same bits as
 * ngacf_score_bwd over the row table followed by ngacf_rows_scatter, one launch.  n = 2B. */
int ngacf_rows_scatter_scored(const int32_t* pos, const float* rows, const float* gscore, void* du, void* di, int dtype,
                              int n, int cu, int d, void* stream);

/* optim.step() of the train loop (train_eval_Gowalla.py:137-138; torch.optim.Adam with L2 weight decay, run_Gowalla.py:114)
 * for up to NGACF_ADAM_MAX_TENSORS fp32 tensors in ONE launch.  The five pointer tables and numel are HOST arrays of `count`
 * entries (copied into the kernel's arguments); steps[t] points at the DEVICE float holding tensor t's step count AFTER this
 * update (the caller increments it first, as torch does): bias corrections are computed from it on the device, so the call
 * can sit in a hipGraph.  Arithmetic of torch's fused Adam: g += wd p; m += (1-b1)(g-m); v = b2 v + (1-b2) g^2;
 * p -= lr/(1-b1^t) * m / (sqrt(v)/sqrt(1-b2^t) + eps).  grads2 (NULL, or a table with NULL entries allowed): a second gradient
 * of the same tensor, g = grads[t] + grads2[t] -- the two propagations of a BPR step (train_eval_Gowalla.py:128-129) produce
 * their parameter gradients separately, and summing them here saves autograd's eight accumulation kernels.  All tensors
 * 16-byte aligned. */
#define NGACF_ADAM_MAX_TENSORS 24
int ngacf_adam_step(float* const* params, const float* const* grads, const float* const* grads2, float* const* exp_avg, float* const* exp_avg_sq,
                    const float* const* steps, const int64_t* numel, int count,
                    float lr, float beta1, float beta2, float eps, float weight_decay, void* stream);

/* The scalar bookkeeping of one train step in one launch: *int_counters[k] += int_strides[k] (device int64: dropout seed
 * counters, the pair sampler's row offset and seed) and *float_counters[k] += 1 (device float: Adam's step counts, which
 * ngacf_adam_step reads).  Host arrays of n_int / n_float (<= NGACF_TICK_MAX) entries. */
#define NGACF_TICK_MAX 48
int ngacf_tick(int64_t* const* int_counters, const int64_t* int_strides, int n_int, float* const* float_counters, int n_float,
               void* stream);

/* BPR loss  -mean(log(sigmoid(pos - neg)))  (graphattention/BPRLoss.py:8-9).
 * loss is a single float; the function overwrites it (no pre-zeroing needed). */
int ngacf_bpr_fwd(const float* pos, const float* neg, float* loss, int B, void* stream);
/* gpos[b] = -gloss * sigmoid(neg-pos) / B ; gneg = -gpos */
int ngacf_bpr_bwd(const float* pos, const float* neg, const float* gloss,
                  float* gpos, float* gneg, int B, void* stream);
/* The loss and its gradients for d loss = 1 in ONE launch (what loss.backward() of train_eval_Gowalla.py:136-137 starts from):
 * loss as ngacf_bpr_fwd, gpos[b] = -sigmoid(neg[b] - pos[b]) / B, gneg = -gpos.  A train step that owns its backward
 * (GraphedBPRStep) feeds these straight to the score nodes: no loss node, no ones-like fill, no second BPR kernel. */
int ngacf_bpr_fwd_bwd(const float* pos, const float* neg, float* loss, float* gpos, float* gneg, int B, void* stream);

/* Static-weight propagation  out[n] = (self_loop ? X[n] : 0) + sum_e val[e] * X[col[e]]
 * (GPLayer.forward, SPUIGACF.py:178-185: torch.sparse.mm(L + I, features)). */
int ngacf_spmm_fwd(const int32_t* segs, int nseg, const int32_t* split, int nsplit,
                   const int32_t* col, const float* val, const void* X, int dtype,
                   void* out, float* part_acc, int N, int d, int self_loop, void* stream);

/* All-item ranking of the evaluation: for each of the B users the K (<= 32) best-scoring candidate items,
 * score(u, i) = F[u] . F[U + i], candidates = items with pool_mask[i] != 0 minus the user's train positives
 * (train_ptr/train_items: CSR over users, items ascending inside a row), best first, ties to the lower item id.
 * Replaces the host-side ranking of train_eval_Gowalla.py:274-354 (per-block score transfer) and :370-385
 * (heapq.nlargest(K, ...) per user in a process pool); the [B][I] score matrix is never materialised.
 * top_items[b][r] = -1 (top_scores = -FLT_MAX) when user b has fewer than r+1 candidates.  d in {64, 128}.
 * Two forms with the same results (bit for bit on tables whose dot products are exact): 64 / 128 / 192 users per workgroup on
 * item rows that a producer wave streams through an LDS ring (128-byte rows always; longer rows from 12 288 users on), or a wave
 * per 32 users that streams the item table itself.  NGACF_TOPK_FORM=lds|wave, NGACF_TOPK_LDS_WAVES=4|8|12 (read per call) force a choice: tests, A/B runs.
 * A users[b] outside [0,U) is never dereferenced: its row comes back all -1 and *bad_index (may be NULL) is bumped. */
int ngacf_eval_topk(const void* F, int dtype, const int64_t* users, int B, int U, int I, int d,
                    const uint8_t* pool_mask, const int64_t* train_ptr, const int32_t* train_items,
                    int K, int32_t* top_items, float* top_scores, int32_t* bad_index, void* stream);

/* BPR pair sampler: for batch row r, u = users[offset + (offset_dev ? *offset_dev : 0) + r]; pos = a uniform draw from
 * the user's train positives (CSR train_ptr/train_items, items ascending); neg = a uniform draw from
 * pool_items[0..P) that is not a train positive of u (rejection, at most 64 rounds; rows that fail bump *unresolved).
 * Replaces data/loadGowalla.py:63-77 (per-user Python sets + random.choice per row).  Randomness: counter hash of
 * (seed + *seed_dev, row, round) -- host restatement: ngacf_amd/rng.py:sample_pairs_host.
 * users has n_rows entries and train_ptr n_users + 1: a batch row past n_rows (a device offset that ran beyond the
 * epoch) or a user id outside [0, n_users) is clamped, never dereferenced, and counted in *unresolved. */
int ngacf_sample_pairs(const int64_t* users, long long n_rows, long long offset, const int64_t* offset_dev,
                       const int64_t* train_ptr, int n_users, const int32_t* train_items,
                       const int32_t* pool_items, int P, uint64_t seed, const uint64_t* seed_dev,
                       int64_t* u_out, int64_t* pos_out, int64_t* neg_out, int* unresolved, int B, void* stream);

/* Dense (optionally masked) multi-head attention, the core of graphattention/modules.py:32-46 `attention` as
 * MultiHeadAttention.forward calls it (:92-107; callers ATTLayer :7-18, ATTLayer_mask :20-30, GACFmodel1.py:29-40,
 * GACFMask.py:96-104).  Q [Nq][dim], K, V [Nk][dim] are the linear layers' outputs as they stand (head h = columns
 * h*dim/heads ..); O [Nq][dim] is the concatenated-heads layout the out projection takes (:104).  Per head:
 *     P = softmax_k( q.k / sqrt(d_k), masked_fill(mask[q][k] == 0, -1e9) );  O = dropout(P) @ v
 * mask: bit-packed, uint32 [Nq][ceil(Nk/32)], bit (k & 31) of word k / 32 set = key k visible to query q; or NULL.  LSE float[Nq][heads][2] = (max, 1 / sum of exp(score - max)) of the masked scaled scores,
 * saved for the backward (kept apart: at a fully masked row max = -1e9 would swallow log(sum) in fp32).
 * Dropout: counter hash of (seed + *seed_dev, head, q, k) like the edge masks.  Exact fp32 matrix cores.  d_k in {8, 16, 32}.
 * The [heads][Nq][Nk] score tensor is never materialised. */
int ngacf_mha_fwd(const float* Q, const float* K, const float* V, const uint32_t* mask, float* O, float* LSE,
                  int Nq, int Nk, int dim, int heads, float drop_p, uint64_t seed, const uint64_t* seed_dev,
                  void* stream);
/* Backward of ngacf_mha_fwd: dQ [Nq][dim], dK, dV [Nk][dim] from dO [Nq][dim]; masked positions pass no gradient
 * (masked_fill).  delta float[Nq][heads] is scratch (dO . O per head).  No atomics: one pass owns query rows, one key rows. */
int ngacf_mha_bwd(const float* Q, const float* K, const float* V, const uint32_t* mask, const float* O,
                  const float* LSE, const float* dO, float* dQ, float* dK, float* dV, float* delta,
                  int Nq, int Nk, int dim, int heads, float drop_p, uint64_t seed, const uint64_t* seed_dev,
                  void* stream);

/* Opt-in kernel timing for measurements (bench.py's roofline leg): between ngacf_timing_begin() and ngacf_timing_end() (any
 * host thread of the process: autograd launches a backward from its own), the launches of the traversal and transform kernels carry start / stop events of their dispatch packets
 * (hipExtLaunchKernel): the kernel's own duration, without the host's launch gaps -- the number a kernel trace reports.
 * ngacf_timing_end synchronises, writes up to cap (duration in ms, NGACF_TAG_*) pairs in launch order and returns their count
 * (negative on error).  Not for use inside a stream capture.  Per-PROCESS state behind a mutex (the launches of any host thread
 * between begin and end are recorded: autograd runs a backward's launches on its own thread); nothing happens unless somebody asks. */
#define NGACF_TAG_PROP_FWD        1   /* prop_fwd_kernel (also with the fused next-stage transform) */
#define NGACF_TAG_PROP_FWD_SPLIT  2   /* prop_fwd_split_kernel (hub rows' partial sums, second kernel) */
#define NGACF_TAG_TRANSFORM_FWD   3   /* transform_mfma_kernel / transform_fwd_kernel, forward */
#define NGACF_TAG_PROP_BWD        4   /* prop_bwd_kernel */
int ngacf_timing_begin(void);
int ngacf_timing_end(float* ms, int32_t* tags, int cap);

/* Peer-store all-gather of the node-partitioned path (stands where the reference has nn.DataParallel's
 * replicate / scatter / gather, parallel.py:94-196): a rank writes its block of a half-table straight into its peers' copies
 * (device pointers mapped through hipIpc) and publishes a counter in each peer's flag word; a consumer waits on its own flag
 * words in front of its traversal.  No collective launch; the epochs live in device memory, so captured graphs replay.
 *   ngacf_peer_alloc / _free      zero-filled device memory owned by the library (hipMalloc): the tables
 *   ngacf_peer_alloc_ctrl         the same, uncached / fine-grained (hipExtMallocWithFlags): the flag words
 *   ngacf_peer_export / _import   64-byte hipIpc handle of such an allocation / the peer's mapping of it (_close unmaps)
 *   ngacf_peer_push   copy every region's src (this rank's block) to dst[0..npeers) (the same block inside each peer's table)
 *                     with system-scope write-through stores; when every workgroup has seen its stores acknowledged, the last
 *                     one (ticket) sets epoch = ++*sent and stores it, system scope, to peer_flags[p] (the word of (slot, this
 *                     rank) in peer p's flag array).  ticket: an int32 of scratch, zero between launches.  bytes % 4 == 0.
 *   ngacf_peer_wait   one wave: until flags[r] has reached ++*expect for every r < n except self_rank, or timeout_ticks of the
 *                     100 MHz clock have passed (*err = 1; the kernel leaves either way).  n <= 64. */
#define NGACF_MAX_PEERS 15
#define NGACF_PEER_MAX_REGIONS 2
#define NGACF_PEER_HANDLE_BYTES 64
typedef struct {
    const void* src;
    uint64_t bytes;
    void* dst[NGACF_MAX_PEERS];
} ngacf_peer_region;
int ngacf_peer_alloc(uint64_t bytes, void** ptr);
/* the same as uncached (fine-grained) device memory: the control block -- flag / epoch words that other agents write while this
 * agent's kernels poll them (coarse-grained memory has no coherence guarantee against remote writes during a kernel) */
int ngacf_peer_alloc_ctrl(uint64_t bytes, void** ptr);
int ngacf_peer_free(void* ptr);
int ngacf_peer_export(const void* ptr, void* handle);
int ngacf_peer_import(const void* handle, void** ptr);
int ngacf_peer_close(void* ptr);
int ngacf_peer_push(const ngacf_peer_region* regions, int nregions, int npeers, int32_t* const* peer_flags,
                    int32_t* sent, int32_t* ticket, void* stream);
int ngacf_peer_wait(const int32_t* flags, int n, int self_rank, int32_t* expect, int32_t* err, uint64_t timeout_ticks,
                    void* stream);

/* Counter-hash dropout mask bits, exposed so tests can pin the device RNG against the
 * host restatement: bits[k] = the 32-bit hash the kernels compare against drop_p * 2^32.
 * kind 0: feature mask, a = node, b = column.  kind 1: edge mask, a = user, b = item (global id), c = head. */
int ngacf_dropout_bits(int kind, uint64_t seed, const int32_t* a, const int32_t* b, const int32_t* c,
                       uint32_t* bits, int n, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* NGACF_H */
