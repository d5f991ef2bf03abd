// pd_ggs_ingest.hip -- asynchronous, device-resident ingestion of pairwise matches (the match container half of
// SURVEY.md section 8f row N2).
//
// Replaces the per-call host prep of util/geometry_guided_sampling.py:16-45 -- the reference re-uploads kp1 / kp2 / i12 on
// EVERY guided step (:19-24), i.e. on its timed path -- and the synchronous host-side pd_ggs_set_matches for callers that
// stream batches: inputs are device-accessible arrays (device memory or pinned host memory) in the reference's own format
// (kp1 / kp2 float64 [M,2], i12 int64 [M,2], demo.py:82-84), CSR-packed over the sequences of a batch; the fp64 -> fp32 cast
// of :167, the pair key of :26-27, the stable sort by pair and every table pd_ggs_kernel reads are built by four small
// kernels on the caller's stream.  No host synchronisation, no allocation in steady state, no host threads.
//
// Stable counting sort by pair key (the order of the matches inside a pair is the upload order, exactly like the host path,
// so sums -- and results -- are bitwise those of pd_ggs_set_matches):
//   ingest_hist     tile of 1024 matches per workgroup: validate, key = i * N + j, LDS histogram -> hist[tile][key]
//   ingest_tables   one workgroup per sequence: per-key totals and per-(tile, key) bases, scan over the keys, compaction
//                   of the non-empty pairs, work items (<= 512 matches), incidence positions per chunk / per frame, and
//                   the sequence descriptor itself (written on the device: the host never learns the counts)
//   ingest_scatter  one wavefront per tile walks its matches IN ORDER, 64 at a time; rank among equal keys of a round
//                   by ballot -> pts[base + rank]
// Because the host never sees the counts, launch shapes are planned from CAPACITIES (pd_match_hints or worst case);
// kernels read the actual counts from the device descriptor.  A violated hint / out-of-range frame index empties the
// slot and raises bit 2 / bit 1 of the engine's async error word (pd_check_async_error).
#include "pd_internal.h"

#include <algorithm>
#include <string.h>
#include <vector>

#define ING_TILE 1024
#define ING_MAX_SEQS 32          // sequences per launch (kernel-argument table); larger batches go in slices
#define ING_TABLE_THREADS 512

struct IngestSeq {
    long long first;             // first row of this sequence in kp1 / kp2 / i12
    int M, n_tiles;
    char *blob;                  // slot blob (layout below)
    PdSeqDesc *desc;             // &eng->d_seqs[slot]
};
struct IngestArgs {
    IngestSeq s[ING_MAX_SEQS];
    const double *kp1, *kp2;
    const long long *i12;
    int N, P_cap, I_cap, C_cap, per_pair_cap;
    int LS_cap;                  // capacity of the lane-major stream in steps per wave (0: no lane-per-item tables for this slice)
    int deg_cap;                 // most pairs one frame may be in (PD_MATCH_HINT_ONE_ORDER: n_frames - 1; else no bound below 2 (n_frames - 1))
    float sc, cx, cy;
    unsigned int *err_flag;
};

// blob layout, by capacity (so that it is known before the counts are)
struct IngestLayout {
    size_t pts, pij, pio, itm, ptb, pco, gps, gio, lit, lwv, lpt, lst, cnt, keys, hist, total;
};
static inline size_t al256(size_t v) { return (v + 255) & ~(size_t)255; }
__host__ __device__ static inline void ingest_layout(int M, int N, int P_cap, int I_cap, int C_cap, int LS_cap, IngestLayout &L) {
    auto al = [](size_t v) { return (v + 255) & ~(size_t)255; };
    const size_t n_tiles = ((size_t)M + ING_TILE - 1) / ING_TILE;
    L.pts = 0;
    L.pij = al(L.pts + sizeof(float4) * (size_t)M);
    L.pio = al(L.pij + sizeof(int2) * (size_t)P_cap);
    L.itm = al(L.pio + sizeof(int) * ((size_t)P_cap + 1));
    L.ptb = al(L.itm + sizeof(int4) * (size_t)I_cap);
    L.pco = al(L.ptb + sizeof(int4) * (size_t)P_cap);
    L.gps = al(L.pco + sizeof(int) * (size_t)C_cap * (N + 1));
    L.gio = al(L.gps + sizeof(int2) * (size_t)P_cap);
    L.lit = al(L.gio + sizeof(int) * ((size_t)N + 1));
    L.lwv = al(L.lit + sizeof(int4) * (size_t)PD_LANE_MAX_ITEMS);
    L.lpt = al(L.lwv + sizeof(int2) * (size_t)PD_LANE_WAVES);
    L.lst = al(L.lpt + sizeof(int2) * (size_t)P_cap);
    L.cnt = al(L.lst + sizeof(float4) * (size_t)LS_cap * PD_LANE_WAVES * 128);
    L.keys = al(L.cnt + sizeof(int) * ((size_t)N * N + 1));
    L.hist = al(L.keys + sizeof(int) * (size_t)M);
    L.total = al(L.hist + sizeof(int) * n_tiles * (size_t)N * N);
}

// ---- kernel 1: keys + per-tile histograms ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void ingest_hist_kernel(IngestArgs A) {
    extern __shared__ int sh_hist[];                  // [N * N]
    const IngestSeq S = A.s[blockIdx.y];
    const int tile = blockIdx.x;
    if (tile >= S.n_tiles) return;
    IngestLayout L;
    ingest_layout(S.M, A.N, A.P_cap, A.I_cap, A.C_cap, A.LS_cap, L);
    const int NN = A.N * A.N;
    for (int q = threadIdx.x; q < NN; q += 256) sh_hist[q] = 0;
    __syncthreads();
    int *keys = (int *)(S.blob + L.keys);
    const int m0 = tile * ING_TILE, m1 = min(S.M, m0 + ING_TILE);
    for (int m = m0 + threadIdx.x; m < m1; m += 256) {
        const long long a = A.i12[2 * (S.first + m)], c = A.i12[2 * (S.first + m) + 1];
        // a frame index out of range: error bit 1, and the match is left out of every count -- the totals then fall short of M, which
        // ingest_tables_kernel treats like a violated hint: the slot is emptied (the synchronous path rejects such an upload outright)
        int key = -1;
        if (a < 0 || a >= A.N || c < 0 || c >= A.N) atomicOr(A.err_flag, 2u);
        else key = (int)(a * A.N + c);                                           // geometry_guided_sampling.py:26-27
        keys[m] = key;
        if (key >= 0) atomicAdd(&sh_hist[key], 1);
    }
    __syncthreads();
    int *hist = (int *)(S.blob + L.hist) + (size_t)tile * NN;
    for (int q = threadIdx.x; q < NN; q += 256) hist[q] = sh_hist[q];
}

// ---- kernel 2: every table of the descriptor ------------------------------------------------------------------------
// inclusive scan of one int per thread over the workgroup (ING_TABLE_THREADS threads); returns the inclusive value, the
// total through `total`.  scratch: [ING_TABLE_THREADS / 64 + 1] ints of LDS.
__device__ __forceinline__ int block_scan_incl(int v, int *scratch, int &total) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int x = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const int y = __shfl_up(x, d, 64);
        if (lane >= d) x += y;
    }
    __syncthreads();
    if (lane == 63) scratch[wave] = x;
    __syncthreads();
    int base = 0, tot = 0;
    for (int w = 0; w < ING_TABLE_THREADS / 64; ++w) {
        const int s = scratch[w];
        if (w < wave) base += s;
        tot += s;
    }
    total = tot;
    return x + base;
}

__global__ __launch_bounds__(ING_TABLE_THREADS) void ingest_tables_kernel(IngestArgs A) {
    extern __shared__ int sh[];
    const IngestSeq S = A.s[blockIdx.x];
    IngestLayout L;
    ingest_layout(S.M, A.N, A.P_cap, A.I_cap, A.C_cap, A.LS_cap, L);
    const int N = A.N, NN = N * N, tid = threadIdx.x;
    // LDS: totals[NN] | pidx[NN] (pair index of a key, -1 if empty) | pair_ij[P_cap] (i | j << 8) | pos0/pos1[P_cap] (shorts
    // packed in one int) | deg / offsets [C_cap + 1][N + 1] | scan scratch [16] | flags [4]
    int *totals = sh;
    int *pidx = totals + NN;
    int *pij = pidx + NN;
    int *ppos = pij + A.P_cap;
    int *foff = ppos + A.P_cap;                     // [(C_cap + 1)][N + 1]
    int *scratch = foff + (A.C_cap + 1) * (N + 1);   // [16]: 0..8 the block scans, 8..15 the block maximum of (3b)
    int *flags = scratch + 16;
    int *lnch = flags + 4, *lstp = lnch + NN;      // (3b) cuts per key, then steps of the key's longest lane item
    int *hist = (int *)(S.blob + L.hist);
    int *cnt = (int *)(S.blob + L.cnt);
    if (tid < 4) flags[tid] = 0;
    // (1) totals per key; hist[tile][key] becomes the exclusive prefix over tiles
    for (int key = tid; key < NN; key += ING_TABLE_THREADS) {
        int run = 0;
        for (int t = 0; t < S.n_tiles; ++t) {
            const int h = hist[(size_t)t * NN + key];
            hist[(size_t)t * NN + key] = run;
            run += h;
        }
        totals[key] = run;
    }
    __syncthreads();
    // (2) scans over the keys, ING_TABLE_THREADS keys per pass: first match of the key, pair index, first work item
    int off_base = 0, pair_base = 0, item_base = 0, multi = 0;
    int4 *items = (int4 *)(S.blob + L.itm);
    int2 *pair_ij_g = (int2 *)(S.blob + L.pij);
    int *pair_item_off = (int *)(S.blob + L.pio);
    int4 *ptab = (int4 *)(S.blob + L.ptb);
    for (int k0 = 0; k0 < NN; k0 += ING_TABLE_THREADS) {
        const int key = k0 + tid;
        const int m = key < NN ? totals[key] : 0;
        const int nch = (m + PD_ITEM_MAX_MATCHES - 1) / PD_ITEM_MAX_MATCHES;
        int t_off, t_pair, t_item;
        const int off = block_scan_incl(m, scratch, t_off) - m + off_base;
        const int p = block_scan_incl(m > 0 ? 1 : 0, scratch, t_pair) - (m > 0 ? 1 : 0) + pair_base;
        const int it = block_scan_incl(nch, scratch, t_item) - nch + item_base;
        if (key < NN) {
            cnt[key] = off;                                   // key_off: first sorted row of the key
            pidx[key] = m > 0 ? p : -1;
            if (m > 0 && p < A.P_cap && it + nch <= A.I_cap) {
                const int i = key / N, j = key - i * N;
                pij[p] = i | (j << 8);
                pair_ij_g[p] = make_int2(i, j);
                pair_item_off[p] = it;
                ptab[p] = make_int4(i | (j << 8), it, nch, 0);
                int start = off;
                for (int c = 0; c < nch; ++c) {               // the host path's split: m / nch (+1 for the first m % nch)
                    const int len = m / nch + (c < m % nch ? 1 : 0);
                    items[it + c] = make_int4(p, start, len, 0);
                    start += len;
                }
                if (nch > 1) multi = 1;
                if (A.per_pair_cap > 0 && m > A.per_pair_cap) atomicOr(&flags[0], 4);
            }
        }
        off_base += t_off;
        pair_base += t_pair;
        item_base += t_item;
    }
    const int n_pairs = pair_base, n_items = item_base;
    if (multi) atomicOr(&flags[1], 1);
    const int n_pchunks = (n_pairs + PD_GGS_THREADS - 1) / PD_GGS_THREADS;
    if (tid == 0) {
        if (n_pairs > A.P_cap || n_items > A.I_cap || n_pchunks > A.C_cap || n_pchunks > PD_GGS_MAX_PCHUNKS || off_base != S.M)
            atomicOr(&flags[0], 4);
    }
    __syncthreads();
    bool bad = flags[0] != 0;                          // (block-uniform; the lane-stream capacity check below may still set it)
    const int np = bad ? 0 : n_pairs;
    if (!bad && tid == 0) pair_item_off[n_pairs] = n_items;
    // (3) incidence positions.  Thread (c, n): c < n_pchunks walks chunk c's pairs for frame n (positions among the chunk's
    // incidences, frame-sorted: ptab.w / pchunk_off); c == C_cap walks ALL pairs (gpos / ginc_off of the two-hop kernel).
    // Pass 1 counts the frame's incidences, a serial scan over the frames gives the offsets, pass 2 assigns the rows.
    for (int pass = 0; pass < 2; ++pass) {
        for (int q = tid; q < (A.C_cap + 1) * N; q += ING_TABLE_THREADS) {
            const int c = q / N, n = q - c * N;
            const bool all = c == A.C_cap;
            if (!all && c >= n_pchunks) {
                if (pass == 0) foff[c * (N + 1) + n] = 0;
                continue;
            }
            const int p_lo = all ? 0 : c * PD_GGS_THREADS, p_hi = all ? np : min(np, p_lo + PD_GGS_THREADS);
            int row = pass == 0 ? 0 : foff[c * (N + 1) + n];
            for (int p = p_lo; p < p_hi; ++p) {
                const int ij = pij[p];
                const int hit0 = (ij & 0xff) == n, hit1 = (ij >> 8) == n;
                if (pass == 1) {
                    if (all) {
                        int2 *gpos = (int2 *)(S.blob + L.gps);
                        if (hit0) gpos[p].x = row;
                        if (hit1) gpos[p].y = row + hit0;
                    } else {
                        // two different threads (frames i and j) write the two halves of ppos[p]: 16-bit stores
                        unsigned short *pp = (unsigned short *)&ppos[p];
                        if (hit0) pp[0] = (unsigned short)row;
                        if (hit1) pp[1] = (unsigned short)(row + hit0);
                    }
                }
                row += hit0 + hit1;
            }
            if (pass == 0) {
                foff[c * (N + 1) + n] = row;                   // degree
                if (row > A.deg_cap) atomicOr(&flags[0], 4);   // a frame in more pairs than the hint allows
            }
        }
        __syncthreads();
        if (pass == 0) bad = flags[0] != 0;                    // (block-uniform again)
        if (pass == 0) {
            if (tid <= A.C_cap) {                              // exclusive scan over the frames of chunk `tid`
                int run = 0;
                for (int n = 0; n < N; ++n) {
                    const int d = foff[tid * (N + 1) + n];
                    foff[tid * (N + 1) + n] = run;
                    run += d;
                }
                foff[tid * (N + 1) + N] = run;
            }
            __syncthreads();
        }
    }
    int *pchunk_off = (int *)(S.blob + L.pco);
    for (int q = tid; q < A.C_cap * (N + 1); q += ING_TABLE_THREADS) pchunk_off[q] = foff[q];
    int *ginc_off = (int *)(S.blob + L.gio);
    for (int q = tid; q <= N; q += ING_TABLE_THREADS) ginc_off[q] = foff[A.C_cap * (N + 1) + q];
    for (int p = tid; p < np; p += ING_TABLE_THREADS) ptab[p].w = ppos[p];
    // (3b) lane-per-item tables (pd_ggs_lane_kernel), exactly as pd_ggs_set_matches builds them: the smallest item length that leaves at
    // most PD_LANE_MAX_ITEMS lane items, balanced cuts of every pair, one lane item per thread, per-wave step counts and stream bases
    int n_litems = 0, n_lwaves = 0, l_len = 0, l_steps = 0;
    int4 *litems = (int4 *)(S.blob + L.lit);
    int2 *lwave = (int2 *)(S.blob + L.lwv);
    int2 *lptab = (int2 *)(S.blob + L.lpt);
    if (A.LS_cap > 0 && !bad && np <= PD_LANE_MAX_ITEMS && n_pchunks == 1 && N <= PD_LANE_MAX_FRAMES) {   // block-uniform
        int mx = 1;
        for (int key = tid; key < NN; key += ING_TABLE_THREADS) mx = max(mx, totals[key]);
        __syncthreads();
        scratch[8 + (tid >> 6)] = 0;
        __syncthreads();
        atomicMax(&scratch[8 + (tid >> 6)], mx);
        __syncthreads();
        int hi = 1;
        for (int w = 0; w < ING_TABLE_THREADS / 64; ++w) hi = max(hi, scratch[8 + w]);
        int lo = 1;
        while (lo < hi) {                                     // block-uniform binary search: every thread takes the same branches
            const int mid = (lo + hi) / 2;
            int part = 0, tot;
            for (int key = tid; key < NN; key += ING_TABLE_THREADS) part += pd_lane_items_of(totals[key], mid);
            block_scan_incl(part, scratch, tot);
            __syncthreads();
            if (tot <= PD_LANE_MAX_ITEMS) hi = mid;
            else lo = mid + 1;
        }
        l_len = lo;
        // cuts per key at that length; the spare lanes go, one more cut each, to the pairs with the longest items; items ordered by the
        // steps of their pair's longest item (descending; key; cut) -- pd_ggs_set_matches' rule, line by line (pd_lane_rank)
        int part = 0, tot0;
        for (int key = tid; key < NN; key += ING_TABLE_THREADS) {
            const int nch = pd_lane_items_of(totals[key], l_len);
            lnch[key] = nch;
            part += nch;
        }
        block_scan_incl(part, scratch, tot0);
        __syncthreads();
        const int spare = PD_LANE_MAX_ITEMS - tot0;
        // round 6: k more cuts for the spare / k pairs with the longest items, k = 1 .. PD_LANE_MORE_MAX by the modelled match pass
        // (pd_lane_pass_cost over the waves' steps) -- pd_ggs_set_matches' loop, line by line
        int *lrk = lstp + NN, *lnk = lrk + NN, *lsk = lnk + NN, *lwt = lsk + NN;      // rank by item length at the base cuts | cuts, steps of a candidate | its waves' longest / shortest item [2][PD_LANE_WAVES]
        for (int key = tid; key < NN; key += ING_TABLE_THREADS) {
            const int m = totals[key], nch = lnch[key];
            lrk[key] = (nch > 0 && m > nch) ? pd_lane_rank(totals, lnch, NN, key, false) : 0;
        }
        __syncthreads();
        int best_cost = 0x7fffffff, best_k = 1, best_d = 0;
        for (int k = 1; k <= PD_LANE_MORE_MAX; ++k)
            for (int d = 0; d < PD_LANE_MORE_SLACK && (d == 0 || spare / k - d > 0); ++d) {        // block-uniform
                int part_k = 0, n_items_k;
                for (int key = tid; key < NN; key += ING_TABLE_THREADS) {
                    const int m = totals[key], nch = lnch[key];
                    const bool elig = nch > 0 && m > nch && lrk[key] < spare / k - d;
                    const int nk = nch + (elig ? min(k, m - nch) : 0);
                    lnk[key] = nk;
                    lsk[key] = nk ? (pd_lane_items_of(m, nk) + 1) / 2 : 0;
                    part_k += nk;
                }
                block_scan_incl(part_k, scratch, n_items_k);
                __syncthreads();
                if (tid < 2 * PD_LANE_WAVES) lwt[tid] = 0;
                __syncthreads();
                for (int key = tid; key < NN; key += ING_TABLE_THREADS) {
                    const int nk = lnk[key];
                    if (!nk) continue;
                    const int first = pd_lane_rank(lsk, lnk, NN, key, true), end = first + nk;
                    for (int w = (first + 63) / 64; w < PD_LANE_WAVES && 64 * w < end; ++w) lwt[w] = lsk[key];
                    for (int w = first / 64; w < PD_LANE_WAVES && 64 * w < end; ++w) {
                        const int last = min(64 * w + 63, n_items_k - 1);
                        if (last < end && last >= first) lwt[PD_LANE_WAVES + w] = lsk[key];
                    }
                }
                __syncthreads();
                int T[PD_LANE_WAVES], Tmin[PD_LANE_WAVES];
                for (int w = 0; w < PD_LANE_WAVES; ++w) {
                    T[w] = lwt[w];
                    Tmin[w] = lwt[PD_LANE_WAVES + w];
                }
                const int cost = pd_lane_pass_cost(T, Tmin);
                if (cost < best_cost) {
                    best_cost = cost;
                    best_k = k;
                    best_d = d;
                }
                __syncthreads();
            }
        for (int key = tid; key < NN; key += ING_TABLE_THREADS) {
            const int m = totals[key], nch = lnch[key];
            lstp[key] = (nch > 0 && m > nch && lrk[key] < spare / best_k - best_d) ? min(best_k, m - nch) : 0;
        }
        __syncthreads();
        part = 0;
        for (int key = tid; key < NN; key += ING_TABLE_THREADS) {
            const int nch = lnch[key] + lstp[key];
            lnch[key] = nch;
            lstp[key] = nch ? (pd_lane_items_of(totals[key], nch) + 1) / 2 : 0;
            part += nch;
        }
        block_scan_incl(part, scratch, n_litems);
        __syncthreads();
        for (int key = tid; key < NN; key += ING_TABLE_THREADS) {
            const int m = totals[key], nch = lnch[key];
            if (m > 0) {
                const int first = pd_lane_rank(lstp, lnch, NN, key, true);
                const int p = pidx[key], i = key / N, j = key - i * N;
                lptab[p] = make_int2(first, nch);
                int start = cnt[key];
                for (int c = 0; c < nch; ++c) {
                    const int len = m / nch + (c < m % nch ? 1 : 0);
                    litems[first + c] = make_int4(i | (j << 8), len, p, start);
                    start += len;
                }
            }
        }
        n_lwaves = (n_litems + 63) / 64;
        __syncthreads();                                       // litems written by this workgroup are read below
        if (tid < n_lwaves) {
            int steps = 0;
            for (int l = 0; l < 64 && tid * 64 + l < n_litems; ++l) steps = max(steps, (litems[tid * 64 + l].y + 1) / 2);
            scratch[tid] = steps;
        }
        __syncthreads();
        int base = 0;
        for (int w = 0; w < n_lwaves; ++w) {
            if (tid == 0) lwave[w] = make_int2(base, scratch[w]);
            base += scratch[w] * 128;
            l_steps = max(l_steps, scratch[w]);
        }
        if (l_steps > A.LS_cap) {                              // cannot happen within the hints; never write past the blob -- and fail
            n_litems = 0;                                      // CLOSED: the slot is emptied and the asynchronous error word raised, like every
            n_lwaves = 0;                                      // other violated hint (the host descriptor still advertises lane tables, so a
            bad = true;                                        // lane-kernel launch must find nothing to do rather than unwritten LDS rows)
        }
        __syncthreads();
    }
    // (4) the descriptor (an emptied slot on error: the GGS kernels then find nothing to do)
    if (tid == 0) {
        PdSeqDesc D;
        D.pts = (const float4 *)(S.blob + L.pts);
        D.pair_ij = (const int2 *)(S.blob + L.pij);
        D.pair_item_off = (const int *)(S.blob + L.pio);
        D.items = (const int4 *)(S.blob + L.itm);
        D.ptab = (const int4 *)(S.blob + L.ptb);
        D.pchunk_off = (const int *)(S.blob + L.pco);
        D.n_pchunks = bad ? 0 : n_pchunks;
        D.gpos = (const int2 *)(S.blob + L.gps);
        D.ginc_off = (const int *)(S.blob + L.gio);
        D.single_item_pairs = flags[1] ? 0 : 1;
        D.lstream = (const float4 *)(S.blob + L.lst);
        D.litems = litems;
        D.lwave = lwave;
        D.lptab = lptab;
        D.n_litems = n_litems;
        D.n_lwaves = n_lwaves;
        D.l_item_len = l_len;
        D.l_max_steps = l_steps;
        D.M = bad ? 1 : S.M;                 // (M only scales 1 / M; never 0: the kernels divide by it)
        D.n_pairs = bad ? 0 : np;
        D.n_items = bad ? 0 : n_items;
        D.n_frames = N;
        D.sc = A.sc;
        D.cx = A.cx;
        D.cy = A.cy;
        D.pad = 0;
        *S.desc = D;
        if (bad) atomicOr(A.err_flag, 4u);
    }
}

// ---- kernel 3: stable scatter ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void ingest_scatter_kernel(IngestArgs A) {
    extern __shared__ int sh_next[];                  // [N * N] next free sorted row of each key for THIS tile
    const IngestSeq S = A.s[blockIdx.y];
    const int tile = blockIdx.x, lane = threadIdx.x;
    if (tile >= S.n_tiles) return;
    IngestLayout L;
    ingest_layout(S.M, A.N, A.P_cap, A.I_cap, A.C_cap, A.LS_cap, L);
    const int NN = A.N * A.N;
    const int *cnt = (const int *)(S.blob + L.cnt);
    const int *hist = (const int *)(S.blob + L.hist) + (size_t)tile * NN;
    for (int q = lane; q < NN; q += 64) sh_next[q] = cnt[q] + hist[q];
    __syncthreads();
    const int *keys = (const int *)(S.blob + L.keys);
    float4 *pts = (float4 *)(S.blob + L.pts);
    const int m0 = tile * ING_TILE, m1 = min(S.M, m0 + ING_TILE);
    for (int r0 = m0; r0 < m1; r0 += 64) {
        const int m = r0 + lane;
        const int key = m < m1 ? keys[m] : -1;
        const bool act = key >= 0;                        // (an out-of-range match has no row: the slot is emptied anyway)
        int dst = -1;
        unsigned long long todo = __builtin_amdgcn_ballot_w64(act);
        while (todo) {                                 // one pass per distinct key of the round (1-2 for pair-grouped input)
            const int leader = __builtin_ctzll(todo);
            const int k = __builtin_amdgcn_readlane(key, leader);
            const unsigned long long same = __builtin_amdgcn_ballot_w64(act && key == k);
            if (act && key == k) dst = sh_next[k] + __builtin_popcountll(same & ((1ull << lane) - 1ull));
            __builtin_amdgcn_wave_barrier();
            if (lane == leader) sh_next[k] += __builtin_popcountll(same);
            __builtin_amdgcn_wave_barrier();
            todo &= ~same;
        }
        if (act) {
            const long long g = S.first + m;
            // .float() of geometry_guided_sampling.py:167 (round-to-nearest fp64 -> fp32), as the host path casts
            pts[dst] = make_float4((float)A.kp1[2 * g], (float)A.kp1[2 * g + 1], (float)A.kp2[2 * g], (float)A.kp2[2 * g + 1]);
        }
    }
}

// ---- kernel 3b: the lane-major stream of pd_ggs_lane_kernel, gathered from the sorted table (before kernel 4 rewrites it) ---------
// one workgroup per (wave of lane items, sequence); thread (l, r) writes lane l's two halves of steps r, r + 4, ...
__global__ __launch_bounds__(256) void ingest_lane_stream_kernel(IngestArgs A) {
    const IngestSeq S = A.s[blockIdx.y];
    const int w = blockIdx.x, l = threadIdx.x & 63;
    const int n_litems = S.desc->n_litems;                // actual counts, written by ingest_tables_kernel
    if (w >= S.desc->n_lwaves) return;
    IngestLayout L;
    ingest_layout(S.M, A.N, A.P_cap, A.I_cap, A.C_cap, A.LS_cap, L);
    const int2 lw = ((const int2 *)(S.blob + L.lwv))[w];
    const float4 *pts = (const float4 *)(S.blob + L.pts);
    float4 *st = (float4 *)(S.blob + L.lst) + lw.x;
    const int q = w * 64 + l;
    const int4 it = q < n_litems ? ((const int4 *)(S.blob + L.lit))[q] : make_int4(0, 0, 0, 0);
    for (int t = threadIdx.x >> 6; t < lw.y; t += 4) {
        float4 q0 = make_float4(1.0f, 1.0f, 1.0f, 1.0f), q1 = q0;
        if (q < n_litems) {
            const float4 a = pts[it.w + min(2 * t, it.y - 1)], b = pts[it.w + min(2 * t + 1, it.y - 1)];
            pd_interleave_pair(a, b, q0, q1);
        }
        st[(2 * t) * 64 + l] = q0;
        st[(2 * t + 1) * 64 + l] = q1;
    }
}

// ---- kernel 4: pair-interleave the full 128-match groups of every item (the layout pd_ggs.hip's packed steps read) -----------
// one wave per work item; a lane rewrites exactly the two elements it read (group[lane], group[64 + lane]): in place, no hazards
__global__ __launch_bounds__(64) void ingest_interleave_kernel(IngestArgs A) {
    const IngestSeq S = A.s[blockIdx.y];
    const int item = blockIdx.x, lane = threadIdx.x;
    if (item >= S.desc->n_items) return;               // actual count, written by ingest_tables_kernel
    IngestLayout L;
    ingest_layout(S.M, A.N, A.P_cap, A.I_cap, A.C_cap, A.LS_cap, L);
    const int4 it = ((const int4 *)(S.blob + L.itm))[item];   // (pair, first match, count, 0)
    float4 *pts = (float4 *)(S.blob + L.pts) + it.y;
    for (int g = 0; g + 128 <= it.z; g += 128) {
        const float4 a = pts[g + lane], b = pts[g + 64 + lane];
        float4 q0, q1;
        pd_interleave_pair(a, b, q0, q1);
        pts[g + lane] = q0;
        pts[g + 64 + lane] = q1;
    }
}

// ---- host side ------------------------------------------------------------------------------------------------------
static size_t tables_lds_bytes(int N, int P_cap, int C_cap) {
    // (+ 3 N^2 for the cut-rule candidates of the lane tables, which exist up to PD_LANE_MAX_FRAMES frames)
    return sizeof(int) * ((size_t)(N <= PD_LANE_MAX_FRAMES ? 7 : 4) * N * N + 2 * PD_LANE_WAVES + 2 * (size_t)P_cap + (size_t)(C_cap + 1) * (N + 1) + 16 + 4);
}

extern "C" int pd_ggs_set_matches_csr_async(pd_engine *eng, int seq_first, int n_seqs, const int64_t *seq_offsets,
                                            const double *kp1, const double *kp2, const int64_t *i12, int n_frames, int height,
                                            int width, const pd_match_hints *hints, void *stream) {
    if (!eng || n_seqs <= 0 || seq_first < 0 || seq_first + n_seqs > eng->max_B || !seq_offsets || !kp1 || !kp2 || !i12) {
        pd_set_error("pd_ggs_set_matches_csr_async: bad engine, slot range [%d, %d) or NULL pointer", seq_first, seq_first + n_seqs);
        return PD_ERR_INVALID_ARG;
    }
    const int N = n_frames;
    if (N <= 0 || N > PD_MAX_FRAMES || N > eng->max_N || height <= 0 || width <= 0) {
        pd_set_error("pd_ggs_set_matches_csr_async: invalid n_frames=%d (<= %d) or image size %dx%d", N, std::min(PD_MAX_FRAMES, eng->max_N),
                     height, width);
        return PD_ERR_INVALID_ARG;
    }
    PD_HIP_CHECK(hipSetDevice(eng->device));
    hipStream_t s = (hipStream_t)stream;
    const bool one_order = hints && hints->max_pairs > 0 && (hints->max_pairs & PD_MATCH_HINT_ONE_ORDER);
    const int hint_pairs = hints ? (hints->max_pairs > 0 ? (hints->max_pairs & ~PD_MATCH_HINT_ONE_ORDER) : hints->max_pairs) : 0;
    const int hint_per_pair = hints ? hints->max_matches_per_pair : 0;
    if (hint_pairs < 0 || hint_per_pair < 0) {
        pd_set_error("pd_ggs_set_matches_csr_async: negative hint");
        return PD_ERR_INVALID_ARG;
    }
    // pass 1: validate every slice and size its capacities -- nothing is enqueued before the whole call is known to be launchable
    struct Slice {
        int nb, P_cap, I_cap, C_cap, LS_cap;
        bool single;
        size_t lds_tab;
    };
    std::vector<Slice> slices;
    for (int b0 = 0; b0 < n_seqs; b0 += ING_MAX_SEQS) {
        Slice sl;
        sl.nb = std::min(ING_MAX_SEQS, n_seqs - b0);
        // one capacity set for the whole slice (the kernels index the layout by it): from the largest sequence
        long long M_max = 0;
        for (int b = 0; b < sl.nb; ++b) {
            const long long M = seq_offsets[b0 + b + 1] - seq_offsets[b0 + b];
            if (M <= 0 || M > 0x7fffffff) {
                pd_set_error("pd_ggs_set_matches_csr_async: sequence %d holds %lld matches (need 1 .. 2^31-1; clear a slot with "
                             "pd_ggs_set_matches(M = 0))", seq_first + b0 + b, M);
                return PD_ERR_INVALID_ARG;
            }
            M_max = std::max(M_max, M);
        }
        sl.P_cap = hint_pairs > 0 ? hint_pairs : (int)std::min<long long>((long long)N * N, M_max);
        sl.single = hint_per_pair > 0 && hint_per_pair <= PD_ITEM_MAX_MATCHES;
        sl.I_cap = sl.single ? sl.P_cap : sl.P_cap + (int)(M_max / PD_ITEM_MAX_MATCHES) + 1;
        sl.C_cap = (sl.P_cap + PD_GGS_THREADS - 1) / PD_GGS_THREADS;
        if (sl.C_cap > PD_GGS_MAX_PCHUNKS) {
            pd_set_error("pd_ggs_set_matches_csr_async: up to %d frame pairs (max %d): pass pd_match_hints.max_pairs", sl.P_cap,
                         PD_GGS_MAX_PCHUNKS * PD_GGS_THREADS);
            return PD_ERR_UNSUPPORTED;
        }
        sl.lds_tab = tables_lds_bytes(N, sl.P_cap, sl.C_cap);
        if (sl.lds_tab > 160 * 1024) {
            pd_set_error("pd_ggs_set_matches_csr_async: tables need %zu B of LDS", sl.lds_tab);
            return PD_ERR_UNSUPPORTED;
        }
        // lane-per-item tables: sum_p ceil(m_p / len) <= M / len + P, so the item length never exceeds ceil(M / (items - P))
        const bool lane_ok = sl.P_cap < PD_LANE_MAX_ITEMS && sl.C_cap == 1 && N <= PD_LANE_MAX_FRAMES;
        const int len_cap = lane_ok ? (int)((M_max + (PD_LANE_MAX_ITEMS - sl.P_cap) - 1) / (PD_LANE_MAX_ITEMS - sl.P_cap)) : 0;
        sl.LS_cap = lane_ok ? (len_cap + 1) / 2 : 0;
        slices.push_back(sl);
    }
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    const bool capturing = hipStreamIsCapturing(s, &cap) == hipSuccess && cap != hipStreamCaptureStatusNone;
    // the engine's own in-flight work (on whatever streams) may still read the slots' tables: a DEVICE-side wait, the host does not
    // block; and earlier uploads on other streams are ordered before this one, so the event recorded at the end covers them too
    if (!capturing) {
        int rc = pd_wait_uses(eng, s, false);
        if (rc) return rc;
        for (auto &e : eng->uploads)
            if (e.stream != s) PD_HIP_CHECK(hipStreamWaitEvent(s, e.event, 0));
    }
    // outgrown blobs whose last readers have finished
    for (size_t i = 0; i < eng->retired_blobs.size();) {
        if (hipEventQuery(eng->retired_blobs[i].done) == hipSuccess) {
            (void)hipFree(eng->retired_blobs[i].ptr);
            (void)hipEventDestroy(eng->retired_blobs[i].done);
            eng->retired_blobs[i] = eng->retired_blobs.back();
            eng->retired_blobs.pop_back();
        } else {
            ++i;
        }
    }
    bool enqueued = false;
    // every exit after the first enqueue records the upload event: later GGS launches on OTHER streams wait for it (pd_sample_phase /
    // pd_ggs_launch), on the device
    auto finish = [&](int rc) {
        if (enqueued && !capturing && pd_record_stream_event(eng->uploads, s) != PD_OK && rc == PD_OK) return (int)PD_ERR_HIP;
        return rc;
    };
    int b0 = 0;
    for (const Slice &sl : slices) {
        const int nb = sl.nb, P_cap = sl.P_cap, I_cap = sl.I_cap, C_cap = sl.C_cap;
        const bool single = sl.single, lane_ok = sl.LS_cap > 0;
        IngestArgs A;
        memset(&A, 0, sizeof(A));
        A.kp1 = kp1;
        A.kp2 = kp2;
        A.i12 = (const long long *)i12;
        A.N = N;
        A.per_pair_cap = hint_per_pair;
        A.sc = (float)std::min(height, width) / 2.0f;   // opencv_from_cameras_projection scale
        A.cx = (float)width / 2.0f;
        A.cy = (float)height / 2.0f;
        A.err_flag = eng->d_err;
        A.P_cap = P_cap;
        A.I_cap = I_cap;
        A.C_cap = C_cap;
        A.LS_cap = sl.LS_cap;
        A.deg_cap = (one_order ? 1 : 2) * (N - 1);
        int max_tiles = 0;
        for (int b = 0; b < nb; ++b) {
            const int slot = seq_first + b0 + b;
            const int M = (int)(seq_offsets[b0 + b + 1] - seq_offsets[b0 + b]);
            IngestLayout L;
            ingest_layout(M, N, P_cap, I_cap, C_cap, A.LS_cap, L);
            PdSeqHost &h = eng->seqs[slot];
            if (h.blob_bytes < L.total) {
                // first use of the slot at this capacity: the only allocation (synchronous).  The old blob may still be read by work
                // in flight: it is parked behind an event on this stream (which has just waited for every use) and freed by a later upload
                if (h.blob) {
                    hipEvent_t ev = nullptr;
                    if (hipEventCreateWithFlags(&ev, hipEventDisableTiming) != hipSuccess || hipEventRecord(ev, s) != hipSuccess) {
                        pd_set_error("pd_ggs_set_matches_csr_async: cannot park the outgrown blob of slot %d", slot);
                        return finish(PD_ERR_HIP);
                    }
                    eng->retired_blobs.push_back({h.blob, ev});
                }
                h.blob = nullptr;
                h.blob_bytes = 0;
                if (hipMalloc(&h.blob, L.total + L.total / 4) != hipSuccess) {   // headroom for ragged batches
                    h.blob = nullptr;
                    memset(&h.desc, 0, sizeof(h.desc));
                    pd_set_error("pd_ggs_set_matches_csr_async: out of device memory for slot %d (%zu B)", slot, L.total + L.total / 4);
                    return finish(PD_ERR_HIP);
                }
                h.blob_bytes = L.total + L.total / 4;
            }
            A.s[b].first = seq_offsets[b0 + b];
            A.s[b].M = M;
            A.s[b].n_tiles = (M + ING_TILE - 1) / ING_TILE;
            A.s[b].blob = (char *)h.blob;
            A.s[b].desc = eng->d_seqs + slot;
            max_tiles = std::max(max_tiles, A.s[b].n_tiles);
            // host shadow = CAPACITIES: launch shapes are planned from these, kernels read the actual counts on the device
            memset(&h.desc, 0, sizeof(h.desc));
            h.desc.M = M;
            h.desc.n_pairs = P_cap;
            h.desc.n_items = I_cap;
            h.desc.n_pchunks = C_cap;
            h.desc.single_item_pairs = single ? 1 : 0;
            h.desc.n_litems = lane_ok ? PD_LANE_MAX_ITEMS : 0;   // capacities again (the plan only asks whether the tables exist)
            h.desc.n_lwaves = lane_ok ? PD_LANE_WAVES : 0;
            h.desc.l_max_steps = A.LS_cap;
            h.desc.n_frames = N;
            h.desc.sc = A.sc;
            h.desc.cx = A.cx;
            h.desc.cy = A.cy;
            h.max_item_len = single ? hint_per_pair : PD_ITEM_MAX_MATCHES;
            // an upper bound (the host never learns the pair list): every pair in both orders, or one (PD_MATCH_HINT_ONE_ORDER; the tables
            // kernel checks the degrees against it)
            h.max_deg = std::min(P_cap, (one_order ? 1 : 2) * (N - 1));
            h.device_built = true;
        }
        const size_t lds_hist = sizeof(int) * (size_t)N * N;
        hipLaunchKernelGGL(ingest_hist_kernel, dim3(max_tiles, nb), dim3(256), lds_hist, s, A);
        enqueued = true;
        hipLaunchKernelGGL(ingest_tables_kernel, dim3(nb), dim3(ING_TABLE_THREADS), sl.lds_tab, s, A);
        hipLaunchKernelGGL(ingest_scatter_kernel, dim3(max_tiles, nb), dim3(64), lds_hist, s, A);
        if (A.LS_cap > 0) hipLaunchKernelGGL(ingest_lane_stream_kernel, dim3(PD_LANE_WAVES, nb), dim3(256), 0, s, A);
        hipLaunchKernelGGL(ingest_interleave_kernel, dim3(I_cap, nb), dim3(64), 0, s, A);
        if (hipGetLastError() != hipSuccess) {
            pd_set_error("pd_ggs_set_matches_csr_async: a table kernel failed to launch");
            return finish(PD_ERR_HIP);
        }
        b0 += nb;
    }
    return finish(PD_OK);
}

int pd_ggs_ingest_init() {
    PD_HIP_CHECK(hipFuncSetAttribute((const void *)ingest_tables_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    return PD_OK;
}
