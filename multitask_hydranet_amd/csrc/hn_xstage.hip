// One chain per XCD: the stride-1 identity XBlocks of a backbone stage as ONE persistent launch (DESIGN.md "persistent stage kernel").
//
// Reference ops: XBlock.forward, net/anynet.py:65-76 (conv_block_1 / 2 / 3 with BatchNorm + ReLU, Squeeze-and-Excitation net/anynet.py:40-48,
// identity shortcut) for blocks 1..d-1 of a stage (net/anynet.py:84-86), training mode.  The launch chain it replaces is ops/backbone.py
// XBlockFn.forward: 8 launches per block, each a dependent trip of ~8-15 us on maps that fit one XCD's L2.
//
// Decomposition.  Workgroup (image i, channel slice s of SL channels), 512 threads, one per CU, all resident at once.  The workgroups of an
// image are placed on ONE XCD -- not assumed from the dispatch order: every workgroup reads HW_REG_XCC_ID and draws a ticket from that XCD's
// counter; ticket t on XCD x works on image x * ipx + t / NS, slice t % NS (surplus workgroups retire).  What crosses XCDs is only the
// BatchNorm statistics (two [SL] vectors per workgroup, three times per block); everything else an XBlock exchanges stays inside the
// image's XCD:
//   phase A  z1 = x W1^T          (workgroup: all HW pixels of its image x its SL couts; W1 slice resident in LDS, x streamed from L2)
//            BN1 statistics       <- all-gather among the N workgroups of the slice (8-byte {tag, value} granules, sc1 both sides)
//            a = relu(bn1(z1))    -> LDS halo tile (the workgroup owns whole images: the 3x3 halo is zero padding, never another workgroup)
//   phase B  z2 = gconv3x3(a)     (compact block form on MFMA: K = 9 taps x 2 groups x 8 channels per 16-cout tile)
//            BN2 statistics       <- all-gather
//            b = relu(bn2(z2)); squeeze; first SE layer as per-slice PARTIAL products, exchanged inside the image (one exchange instead of
//            squeeze + hidden); second SE layer for the slice's own channels; bg = b * gate -> global (operand of conv_block_3)
//   phase C  z3 = bg W3^T; BN3 statistics <- all-gather; out = relu(bn3(z3) + x) -> global (operand of the next block)
// The backward launch (hn_xstage_bwd) walks the same blocks in reverse with the same decomposition: BatchNorm-backward sums cross XCDs,
// dz3 / dz1 (operands of the two data-gradient GEMMs) and the SE partials stay in the image's XCD, and a block's input gradient dx is the
// next block's output gradient of the SAME workgroup (same image, same channel slice): it never leaves the registers.
// Every tensor the launch chain's backward reads (z1, a, z2, bg, z3, out, BatchNorm coefficients, pooled / hidden / gate vectors, running
// statistics) is written exactly as XBlockFn.forward leaves it, so either backward can follow.
//
// Synchronisation.  In-image exchanges (bg, out, SE partials): plain stores, every storing wave's vmcnt(0), workgroup barrier, one arrival
// on a per-image counter; consumers poll it and then read through the XCD's L2 (activation tensors are written once per launch, so no CU's
// L1 can hold an older copy; the small vectors are read with sc1 loads).  LOCAL = true: the counters are L2-resident (workgroup-scope
// read-modify-writes execute in the shared L2: 1.0 us per round measured, profiles/r05_xcd_sync_probe.txt) -- valid because co-location was
// READ from the hardware, not assumed; LOCAL = false: agent-scope counters, release / acquire fences around the payload (placement
// independent, slower; kept as the checked fallback form).  Cross-XCD: granules only.  Every spin is bounded (XS_TIMEOUT_TICKS of the 100 MHz
// real-time counter); on expiry the workgroup raises the status word, everyone else sees it in its own polls and retires: a launch that
// cannot become resident ends with status != 0 instead of hanging.  State words are reset by the last workgroup to leave (the granule tags
// are epochs that continue across launches: no memset node, replay-safe).
#include "hn_common.h"

#define XS_MAXB 16
#define XS_THREADS 512
#define XS_GRID 256
#define XS_MAX_IMG 32
#define XS_MAX_SLICES 16
#define XS_TIMEOUT_TICKS 20000000ull         // 200 ms of the 100 MHz real-time counter (a collective on a side stream may hold CUs for milliseconds)
#define XS_CNT_OFF 4096
#define XS_HP_OFF (XS_CNT_OFF + XS_MAX_IMG * 4 * 128)
#define XS_GRAN_OFF (XS_HP_OFF + XS_MAX_IMG * XS_MAX_SLICES * 256 * 4)
#define XS_DBG_OFF (XS_GRAN_OFF + 2 * XS_MAX_SLICES * XS_MAX_IMG * 128 * 8)      // [XS_GRID][4] uint32: xcc << 16 | ticket, failure code, block
#define XS_WS_BYTES (XS_DBG_OFF + XS_GRID * 16)

typedef __attribute__((address_space(1))) unsigned gu32;
typedef __attribute__((address_space(1))) unsigned long long gu64;
typedef __attribute__((address_space(1))) float gf32;

struct XsBlock {
    const bf16* w1;                    // conv_block_1 weight, packed bf16 [C][KP]                       (backward: the transposed pack wt1)
    const bf16* w2;                    // conv_block_2 weight, block-diagonal pack [C][9][64] (hn_gconv_pack_diag wk; backward: wd)
    const bf16* w3;                    // conv_block_3 weight, packed bf16 [C][KP]                       (backward: wt3)
    const float* sw1; const float* sb1; const float* sw2; const float* sb2;   // SE: [Cs][C], [Cs], [C][Cs], [C]
    const float* g1; const float* b1; float* rm1; float* rv1;
    const float* g2; const float* b2; float* rm2; float* rv2;
    const float* g3; const float* b3; float* rm3; float* rv3;
};
struct XsCommon {
    int N, H, W, C, KP, Cs;
    char* ws;
    int ipx, NS;
    unsigned long long* stamps;        // optional [nb][16] real-time stamps of workgroup (image 0, slice 0)
    int dbg;                           // timing experiments (tools/): 2 = no pixel-operand loads behind the first stage, 4 = no LDS reads / MFMAs,
                                       // 8 = no weight-slice loads, 16 = both GEMMs stream the launch's input x0 instead of out / bg
};
struct XsArgs {
    XsBlock blk[XS_MAXB];
    int nb;
    XsCommon g;
    const bf16* x0;                    // input of the first block [N * HW][C]
    bf16 *z1, *a, *z2, *bg, *z3, *out; // [nb][N * HW][C]
    float* coef;                       // [nb][3][4][C]: scale, shift, mean, rstd of BatchNorm 1 / 2 / 3
    float *pooled, *hid, *gate;        // [nb][N][C], [nb][N][Cs], [nb][N][C]
    float eps, momentum, alpha;        // alpha = 1 / (H * W) as the launch chain passes it
};
// backward: per block the three transposed / flipped weight packs and the SE weights (the forward's saved tensors travel as stacked arrays)
struct XbBlock { const bf16* wt1; const bf16* wd2; const bf16* wt3; const float* sw1; const float* sw2; };
struct XbArgs {
    XbBlock blk[XS_MAXB];
    int nb;
    XsCommon g;
    const bf16* dout;                  // gradient of the LAST block's output [N * HW][C]
    const bf16 *z1, *z2, *z3, *out;    // [nb][N * HW][C] saved by the forward
    const float* coef;                 // [nb][3][4][C]
    const float *hid, *gate;           // [nb][N][Cs], [nb][N][C]
    bf16 *dz1, *dz2, *dz3;             // [nb][N * HW][C]: operands of the deferred weight gradients
    bf16* dx;                          // [N * HW][C]: gradient of the first block's input
    float* dgb;                        // [nb][3][2][C]: (dgamma, dbeta) of BatchNorm 1 / 2 / 3
    float *dpre2, *dpre1;              // [nb][N][C], [nb][N][Cs]: pre-activation gradients of the SE layers (deferred outer products)
};

__device__ __attribute__((aligned(16))) bf16 xs_zero_piece[8];

__device__ __forceinline__ void xs_glds16(const bf16* src, char* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src, (__attribute__((address_space(3))) void*)lds_wave_base,
                                     16, 0, 0);
}
// 128-byte LDS rows (64 k), 16-byte pieces XOR-swizzled by the row (hn_gemm.hip swz): conflict-free for ds_read_b128 with lane & 15 = row
__device__ __forceinline__ int xs_swz(int row, int piece) { return row * 128 + ((piece ^ (row & 7)) << 4); }
__device__ __forceinline__ unsigned xs_xcc_id() { return __builtin_amdgcn_s_getreg((3 << 11) | 20) & 0xfu; }     // HW_REG_XCC_ID[3:0]
__device__ __forceinline__ unsigned long long xs_now() { return __builtin_amdgcn_s_memrealtime(); }

template <bool LOCAL>
__device__ __forceinline__ void xs_arrive(gu32* ctr) {
    if (LOCAL) __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    else __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ unsigned xs_peek(gu32* ctr) {
    // an sc1 load: bypasses this CU's vector cache and is served by the L2 the arrivals execute in.  (NOT a workgroup-scope fetch_add of
    // zero: hipcc turns the idempotent read-modify-write into a `global_load_dword sc0`, which the vector cache may serve for ever.)
    return __hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// one lane: wait until *ctr >= target; false on expiry or when another workgroup raised the status word
__device__ __forceinline__ bool xs_wait(gu32* ctr, unsigned target, gu32* status) {
    const unsigned long long t0 = xs_now();
    for (unsigned n = 1;; ++n) {
        if ((int)(xs_peek(ctr) - target) >= 0) return true;
        __builtin_amdgcn_s_sleep(1);
        if ((n & 63u) == 0 && (__hip_atomic_load(status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0 || xs_now() - t0 > XS_TIMEOUT_TICKS))
            return false;
    }
}

// ---------------------------------------------------------------------------------------------------------------------------------
// The workgroup: its place (image, slice), its LDS, and the operations both launches are made of.  Waves 0-3 store the tiles other
// workgroups stream and sweep the granules; waves 4-7 issue the weight-slice DMAs, touch lines and store the tensors nobody waits for
// inside the launch -- vmcnt completes in order per wave, so a latency-critical wait must not sit behind a bulk prefetch of its own wave.
// ---------------------------------------------------------------------------------------------------------------------------------
template <int SL, int HWP, bool LOCAL>
struct XsWg {
    static constexpr int WGC = SL / 32, WGP = 8 / WGC, TC = 2, WP = HWP / WGP, TP = WP / 16;
    static constexpr int XSTAGE = HWP * 128;             // one 64-deep K stage of the pixel operand: whole 128-byte segments per row (64-byte
                                                         // segments stream at 29 GB/s per CU whatever the depth, 128-byte ones at 65-72 with ONE
                                                         // stage in flight and slower with three: tools/xstage/feed_probe.hip)
    static constexpr int NI = HWP / 64;                  // LDS-DMA instructions per wave and stage
    static constexpr int RS = SL * 2 + 16;               // row stride of the staging / halo tiles (bytes): 16 consecutive rows hit 16 bank groups
    static constexpr int V2 = 2 * SL;                    // statistics values per workgroup

    const XsCommon& g;
    int tid, lane, wave, wc, wp;
    int C, KP, Cs, HW, W2;
    char *Wreg, *Ring, *Misc, *Stg;
    float *gath, *red, *csc, *csh, *lpool, *lh, *lgate, *lcoef, *lvec;
    int* lflag;
    gu32 *ctl, *status, *cnt;
    unsigned* dbg;
    int img, slice, c0, SLv;
    unsigned ebase;
    bool working, dead;
    int cur_b;
    unsigned touched;
    f32x4 acc[TC][TP];
    float q[TC][TP][4];

    __device__ __forceinline__ XsWg(const XsCommon& g_, char* smem) : g(g_) {
        tid = threadIdx.x;
        ids();
        C = g.C; KP = g.KP; Cs = g.Cs; HW = g.H * g.W; W2 = g.W + 2;
        Wreg = smem;
        Ring = smem + SL * ((KP + 63) >> 6) * 128;
        Misc = Ring + 2 * XSTAGE;
        Stg = Ring;                                                      // [HWP][RS] output staging tile / [(H+2)(W+2)][RS] halo tile
        gath = reinterpret_cast<float*>(Ring + HWP * RS);                // [N][V2] gathered statistics
        red = reinterpret_cast<float*>(Misc);                            // [WGP][2][SL]
        csc = reinterpret_cast<float*>(Misc + 2048);                     // [SL] scale (backward: mean of g)
        csh = csc + SL;                                                  // [SL] shift (backward: mean of g * xhat)
        lpool = reinterpret_cast<float*>(Misc + 2560);                   // [SL]
        lh = reinterpret_cast<float*>(Misc + 2816);                      // [256]
        lgate = reinterpret_cast<float*>(Misc + 3840);                   // [SL]
        lflag = reinterpret_cast<int*>(Misc + 4096);                     // ticket, epoch base, dead, xcc
        lcoef = reinterpret_cast<float*>(Misc + 4352);                   // [4][SL] forward coefficients of the BatchNorm in hand (backward)
        lvec = reinterpret_cast<float*>(Misc + 5376);                    // [2][256] scratch vectors (backward SE)
        ctl = (gu32*)(g.ws);
        status = ctl + 64;
        dbg = reinterpret_cast<unsigned*>(g.ws + XS_DBG_OFF) + blockIdx.x * 4;
        if (tid == 0) {
            const unsigned xcc = xs_xcc_id();
            lflag[3] = (int)xcc;
            lflag[0] = (int)__hip_atomic_fetch_add(ctl + 96 + 32 * (xcc & 7u), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            lflag[1] = (int)__hip_atomic_load(ctl, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            lflag[2] = 0;
            dbg[0] = (xcc << 16) | (unsigned)lflag[0];
            dbg[1] = 0;
        }
        __syncthreads();
        const int ticket = lflag[0], xcc = lflag[3];
        ebase = (unsigned)lflag[1];
        img = xcc * g.ipx + ticket / g.NS;
        slice = ticket % g.NS;
        working = xcc < 8 && ticket < g.ipx * g.NS && img < g.N;
        c0 = slice * SL;
        SLv = working ? (C - c0 < SL ? C - c0 : SL) : 0;                 // real channels of the slice (a multiple of 8)
        cnt = (gu32*)(g.ws + XS_CNT_OFF) + (long)img * 128;              // [kind 0, 1, 2, 3 = leave][32]
        dead = false;
        cur_b = 0;
        touched = 0;
    }
    // (the wave index as a scalar: derived from the opaque tid it lives in a vector register, and so does every LDS-DMA destination, M0
    // value and row base computed from it -- the wide backward kernel spilled those and reloaded them in front of every stage's requests)
    __device__ __forceinline__ void ids() { lane = tid & 63; wave = __builtin_amdgcn_readfirstlane(tid >> 6); wc = wave / WGP; wp = wave % WGP; }
    // (opaque per block: per-lane address arithmetic is recomputed where it is used; hoisted out of the block loop it was ~55 spilled
    // 64-bit values whose reloads queue behind the prefetches in the in-order vmcnt stream)
    __device__ __forceinline__ void refresh() { asm volatile("" : "+v"(tid)); ids(); }
    __device__ __forceinline__ void stamp(int b, int k) const {
        if (g.stamps && img == 0 && slice == 0 && tid == 0) g.stamps[b * 16 + k] = xs_now();
    }
    // all of this workgroup's stores are in the L2 (or, !LOCAL, written back) -> one arrival on the image's counter
    // (waves 4-7 issue the weight-slice DMAs and never store a streamed tile: `stores_only` leaves their DMAs in flight)
    __device__ __forceinline__ void arrive(int kind, bool stores_only = false) {
        if (!stores_only || wave < 4) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0) {
            if (!LOCAL) {
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            xs_arrive<LOCAL>(cnt + kind * 32);
        }
    }
    __device__ __forceinline__ void await(int kind, unsigned target) {
        if (tid == 0) {
            if (!xs_wait(cnt + kind * 32, target, status)) {
                lflag[2] = 1;
                __hip_atomic_fetch_or(status, 0x100u | (unsigned)kind, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                dbg[1] = 0x100u | (unsigned)kind; dbg[2] = (unsigned)cur_b; dbg[3] = xs_peek(cnt + kind * 32);
            }
            if (!LOCAL) {
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
        }
        __syncthreads();
        dead = lflag[2] != 0;
    }
    // weight slice [SL rows c0..][KP] -> LDS, 64-deep chunks of [SL][128 B], pieces swizzled on the source side (waves 4-7)
    __device__ __forceinline__ void issue_w(const bf16* Wp) {
        if (g.dbg & 8) return;
        constexpr int RB = SL / 8;                       // LDS-DMA instructions (8 rows x 8 pieces) per chunk
        const int nq = ((KP + 63) >> 6) * RB;
        if (wave < 4) return;
        for (int qq = wave - 4; qq < nq; qq += 4) {
            const int chunk = qq / RB, rb = qq - chunk * RB;
            const int row = rb * 8 + (lane >> 3);
            const int k = chunk * 64 + (((lane & 7) ^ (row & 7)) << 3);
            const int co = c0 + row;
            const bf16* src = (co < C && k < KP) ? Wp + (long)co * KP + k : xs_zero_piece;
            xs_glds16(src, Wreg + qq * 1024);
        }
    }
    // acc = X[image] (HW x C, zero rows behind HW) * Wreg^T: double-buffered 64-deep K stages, one barrier per stage
    __device__ __forceinline__ void gemm(const bf16* X) {
        const int S = (KP + 63) >> 6;
        auto issue_x = [&](int s) {
            char* dst = Ring + (s & 1) * XSTAGE;
#pragma unroll
            for (int u = 0; u < NI; ++u) {
                const int rb = u * 8 + wave;
                const int row = rb * 8 + (lane >> 3);
                const int k = s * 64 + (((lane & 7) ^ (row & 7)) << 3);
                const bf16* src = (row < HW && k < C) ? X + (long)row * C + k : xs_zero_piece;
                xs_glds16(src, dst + rb * 1024);
            }
        };
#pragma unroll
        for (int i = 0; i < TC; ++i)
#pragma unroll
            for (int j = 0; j < TP; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
        issue_x(0);
        for (int it = 0; it < S; ++it) {
            asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");     // stage `it` (and the weight slice) landed; slot (it+1)&1 is free
            if (it + 1 < S && !(g.dbg & 2)) issue_x(it + 1);
            if (g.dbg & 4) continue;
            const char* sX = Ring + (it & 1) * XSTAGE;
            const char* sW = Wreg + it * (SL * 128);
            if constexpr (TP <= 2) {                       // narrow tiles: all fragment reads of the stage first, then the MFMAs
                bf16x8 fa[2][TC], fb[2][TP];
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) {
                    const int piece = ks * 4 + (lane >> 4);
#pragma unroll
                    for (int i = 0; i < TC; ++i) fa[ks][i] = *reinterpret_cast<const bf16x8*>(sW + xs_swz(wc * 32 + i * 16 + (lane & 15), piece));
#pragma unroll
                    for (int j = 0; j < TP; ++j) fb[ks][j] = *reinterpret_cast<const bf16x8*>(sX + xs_swz(wp * WP + j * 16 + (lane & 15), piece));
                }
#pragma unroll
                for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                    for (int i = 0; i < TC; ++i)
#pragma unroll
                        for (int j = 0; j < TP; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[ks][i], fb[ks][j], acc[i][j], 0, 0, 0);
            } else {                                       // wide tiles: eight MFMAs per half stage cover the next half's reads (24 registers less)
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) {
                    bf16x8 fa[TC], fb[TP];
                    const int piece = ks * 4 + (lane >> 4);
#pragma unroll
                    for (int i = 0; i < TC; ++i) fa[i] = *reinterpret_cast<const bf16x8*>(sW + xs_swz(wc * 32 + i * 16 + (lane & 15), piece));
#pragma unroll
                    for (int j = 0; j < TP; ++j) fb[j] = *reinterpret_cast<const bf16x8*>(sX + xs_swz(wp * WP + j * 16 + (lane & 15), piece));
#pragma unroll
                    for (int i = 0; i < TC; ++i)
#pragma unroll
                        for (int j = 0; j < TP; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[i], fb[j], acc[i][j], 0, 0, 0);
                }
            }
        }
        __syncthreads();                                 // every wave is done with the ring and the weight slice
    }
    // grouped 3x3 conv of the halo tile (compact block form): per 16-cout tile the K index of chunk jc, piece pq is (tap = e >> 1, group half =
    // e & 1) with e = 4 jc + pq; a cout's operand row is non-zero on its own group only.  load_wf: the operand fragments from a
    // block-diagonal pack [C][9][64] (hn_gconv_pack_diag: wk forward, wd data gradient)
    __device__ __forceinline__ void load_wf(const bf16* w2, bf16x8 (&wf)[TC][5]) const {
#pragma unroll
        for (int i = 0; i < TC; ++i) {
            const int co = c0 + wc * 32 + i * 16 + (lane & 15);
#pragma unroll
            for (int jc = 0; jc < 5; ++jc) {
                const int e = 4 * jc + (lane >> 4), t = e >> 1, h = e & 1;
                const bool v = e < 18 && co < C && ((co >> 3) & 1) == h;
                wf[i][jc] = v ? ld8(w2 + (long)co * 576 + t * 64 + ((co & 63) >> 3) * 8) : zero8();
            }
        }
    }
    __device__ __forceinline__ void gconv(const bf16x8 (&wf)[TC][5]) {
#pragma unroll
        for (int i = 0; i < TC; ++i)
#pragma unroll
            for (int j = 0; j < TP; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
        int hb[TP];
        bool hv[TP];
#pragma unroll
        for (int j = 0; j < TP; ++j) {
            const int px = px_of(j);
            hv[j] = px < HW;
            const int y = hv[j] ? px / g.W : 0, x = hv[j] ? px - y * g.W : 0;
            hb[j] = (y * W2 + x) * RS;
        }
#pragma unroll
        for (int jc = 0; jc < 5; ++jc) {
            const int e = 4 * jc + (lane >> 4), t = e >> 1, h = e & 1;
            const int ky = (t * 11) >> 5, kx = t - 3 * ky;
            const int toff = (ky * W2 + kx) * RS + h * 16;
#pragma unroll
            for (int i = 0; i < TC; ++i) {
                const int choff = (wc * 32 + i * 16) * 2;
#pragma unroll
                for (int j = 0; j < TP; ++j) {
                    const bf16x8 fb = (e < 18 && hv[j]) ? *reinterpret_cast<const bf16x8*>(Stg + hb[j] + toff + choff) : zero8();
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[i][jc], fb, acc[i][j], 0, 0, 0);
                }
            }
        }
    }
    // per-lane coordinates of the accumulator tiles: couts co(i) .. co(i)+3, pixel px(j)
    __device__ __forceinline__ int co_of(int i) const { return wc * 32 + i * 16 + (lane >> 4) * 4; }
    __device__ __forceinline__ int px_of(int j) const { return wp * WP + j * 16 + (lane & 15); }
    __device__ __forceinline__ void acc_to_q(bool mask_px) {
#pragma unroll
        for (int i = 0; i < TC; ++i)
#pragma unroll
            for (int j = 0; j < TP; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) q[i][j][r] = (!mask_px || px_of(j) < HW) ? bfround(acc[i][j][r]) : 0.f;
    }
    // per-channel sums over the workgroup's pixels into red[wp][0 / 1][c]: kind 0: (sum q, sum q^2), 1: sum q only,
    // 2: (sum q, sum q * o[..]) with a second per-element operand
    template <int KIND>
    __device__ __forceinline__ void tile_sums(const float (*o)[TP][4] = nullptr) {
#pragma unroll
        for (int i = 0; i < TC; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float s1 = 0.f, s2 = 0.f;
#pragma unroll
                for (int j = 0; j < TP; ++j) {
                    s1 += q[i][j][r];
                    if (KIND == 0) s2 += q[i][j][r] * q[i][j][r];
                    if (KIND == 2) s2 += q[i][j][r] * o[i][j][r];
                }
                s1 = row16_sum(s1);
                if (KIND != 1) s2 = row16_sum(s2);
                if ((lane & 15) == 0) {
                    red[(wp * 2 + 0) * SL + co_of(i) + r] = s1;
                    if (KIND != 1) red[(wp * 2 + 1) * SL + co_of(i) + r] = s2;
                }
            }
    }
    __device__ __forceinline__ void zero_halo() {
        for (int i = tid * 16; i < (g.H + 2) * W2 * RS; i += XS_THREADS * 16) *reinterpret_cast<u32x4*>(Stg + i) = (u32x4){0u, 0u, 0u, 0u};
    }
    __device__ __forceinline__ void stage_tile(bool halo) {     // q -> bf16 staging tile (plain [px] rows, or the interior of the halo tile)
#pragma unroll
        for (int j = 0; j < TP; ++j) {
            const int px = px_of(j);
            if (halo && px >= HW) continue;
            const int y = halo ? px / g.W : 0, x = halo ? px - y * g.W : 0;
            const int row = halo ? (y + 1) * W2 + x + 1 : px;
#pragma unroll
            for (int i = 0; i < TC; ++i) {
                bf16x4 v;
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = f2bf(q[i][j][r]);
                *reinterpret_cast<bf16x4*>(Stg + row * RS + co_of(i) * 2) = v;
            }
        }
    }
    // staging tile -> the image's rows of a [N * HW][C] tensor, channels [c0, c0 + SLv).  Tiles another workgroup streams: waves 0-3, whose
    // queue arrive() drains; tensors nobody waits for inside the launch (hi): waves 4-7, behind their DMAs
    __device__ __forceinline__ void store_tile(bf16* dst, bool halo, bool hi = true) const {
        const int PR = SLv >> 3, total = HW * PR;
        bf16* base = dst + (long)img * HW * C + c0;
        if ((wave >= 4) != hi) return;
        for (int idx = tid & 255; idx < total; idx += 256) {
            const int px = idx / PR, pc = idx - px * PR;
            int row = px;
            if (halo) { const int y = px / g.W, x = px - y * g.W; row = (y + 1) * W2 + x + 1; }
            st8(base + (long)px * C + pc * 8, *reinterpret_cast<const bf16x8*>(Stg + row * RS + pc * 16));
        }
    }
    // this lane's pieces of the workgroup's tile of a [N * HW][C] tensor, in the accumulator layout (4 channels x 1 pixel per (i, j))
    __device__ __forceinline__ void load_tile(const bf16* src, bf16x4 (&t)[TC][TP]) const {
#pragma unroll
        for (int i = 0; i < TC; ++i)
#pragma unroll
            for (int j = 0; j < TP; ++j) {
                const int px = px_of(j), co = co_of(i);
                t[i][j] = (px < HW && co < SLv) ? *reinterpret_cast<const bf16x4*>(src + ((long)img * HW + px) * C + c0 + co) : (bf16x4){};
            }
    }
    // Lines whose FIRST touch is a write are not served from the L2 afterwards (tools/xstage/feed_probe.hip: the image's 240 KB stream in
    // 6.3 us from a tile written to untouched lines, in 3.7 us when the lines had been read before): the tile a GEMM of the other workgroups
    // will stream is touched (one dword per 64 bytes, waves 4-7) a few microseconds before it is written.
    __device__ __forceinline__ void touch_tile(const bf16* dst) {
        const int t2 = tid & 255;
        if (wave < 4) return;
        for (int r = t2 >> 1; r < HW; r += 128) {
            const int off = (t2 & 1) * 32;
            if (off < SLv) touched += *reinterpret_cast<const unsigned*>(dst + ((long)img * HW + r) * C + c0 + off);
        }
    }
    // Statistics of the slice over ALL images, in two steps so that other trips run under the wait: publish (this workgroup's two sums per
    // channel, from red, leave as tagged granules); gather (the N workgroups of the slice are collected into gath by waves 0-3).
    // e = exchange index of the launch.
    __device__ __forceinline__ gu64* gran_of(int e) const {
        return (gu64*)(g.ws + XS_GRAN_OFF) + ((long)((e & 1) * XS_MAX_SLICES + slice) * XS_MAX_IMG) * 128;
    }
    __device__ __forceinline__ void publish(int e) {
        if (tid < V2) {
            const int k = tid / SL, c = tid - k * SL;
            float v = 0.f;
#pragma unroll
            for (int w = 0; w < WGP; ++w) v += red[(w * 2 + k) * SL + c];
            __hip_atomic_store(gran_of(e) + (long)img * 128 + tid, ((unsigned long long)(ebase + (unsigned)e + 1u) << 32) | __float_as_uint(v),
                               __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    __device__ __forceinline__ void gather(int e) {
        const unsigned tag = ebase + (unsigned)e + 1u;
        gu64* const gr = gran_of(e);
        const int total = g.N * V2;
        const unsigned long long t0 = xs_now();
        bool ok = true;
        // every pass requests all of the thread's granules at once (one trip per pass, not one per granule)
        constexpr int NGM = 8;                            // granules per thread of the sweeping waves 0-3: N * V2 <= 2048 (checked on the host)
        unsigned long long gx[NGM];
        for (unsigned n = 1; wave < 4; ++n) {
            bool all = true;
#pragma unroll
            for (int k = 0; k < NGM; ++k) {
                const int idx = tid + k * 256;
                gx[k] = idx < total ? __hip_atomic_load(gr + (long)(idx / V2) * 128 + (idx % V2), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
                                    : ((unsigned long long)tag << 32);
            }
#pragma unroll
            for (int k = 0; k < NGM; ++k) all = all && (unsigned)(gx[k] >> 32) == tag;
            if (all) break;
            __builtin_amdgcn_s_sleep(1);
            if ((n & 63u) == 0 && (__hip_atomic_load(status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0 || xs_now() - t0 > XS_TIMEOUT_TICKS)) {
                ok = false;
                break;
            }
        }
#pragma unroll
        for (int k = 0; k < NGM; ++k) {
            const int idx = tid + k * 256;
            if (wave < 4 && idx < total) gath[idx] = __uint_as_float((unsigned)gx[k]);
        }
        if (!ok) {
            lflag[2] = 1;
            __hip_atomic_fetch_or(status, 0x200u | (unsigned)(e & 0xff), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            dbg[1] = 0x200u | (unsigned)(e & 0xff); dbg[2] = (unsigned)cur_b; dbg[3] = (unsigned)tid;
        }
        __syncthreads();
        dead = lflag[2] != 0;
    }
    // totals of channel `c` (tid) over the N images, in double (the launch chain folds its fp32 partial rows the same way)
    __device__ __forceinline__ void totals(int c, double& s1, double& s2) const {
        s1 = 0.0; s2 = 0.0;
        for (int im = 0; im < g.N; ++im) { s1 += (double)gath[im * V2 + c]; s2 += (double)gath[im * V2 + SL + c]; }
    }
    // forward BatchNorm: coefficients into csc / csh; the image-0 workgroup stores them and updates the running statistics
    __device__ __forceinline__ void bn_forward(int e, const float* gamma, const float* beta, float* rm, float* rv, float* coef_out, float eps,
                                               float momentum) {
        float ga = 0.f, be = 0.f, rm0 = 0.f, rv0 = 0.f;
        if (tid < SLv) {                                  // requested in front of the gather: behind it they were a trip of their own
            ga = gamma[c0 + tid]; be = beta[c0 + tid];
            if (img == 0) { rm0 = rm[c0 + tid]; rv0 = rv[c0 + tid]; }
        }
        gather(e);
        if (tid < SL) {
            float sc = 0.f, sh = 0.f;
            if (tid < SLv && !dead) {
                double s1, s2;
                totals(tid, s1, s2);
                const double count = (double)g.N * HW;
                const double mu = s1 / count;
                double var = s2 / count - mu * mu;
                if (var < 0.0) var = 0.0;
                const float rs = (float)(1.0 / sqrt(var + (double)eps));
                sc = ga * rs;
                sh = be - (float)mu * sc;
                if (img == 0) {
                    const int c = c0 + tid;
                    coef_out[c] = sc; coef_out[C + c] = sh; coef_out[2 * C + c] = (float)mu; coef_out[3 * C + c] = rs;
                    const double unb = count > 1.0 ? var * count / (count - 1.0) : var;
                    rm[c] = (1.f - momentum) * rm0 + momentum * (float)mu;
                    rv[c] = (1.f - momentum) * rv0 + momentum * (float)unb;
                }
            }
            csc[tid] = sc; csh[tid] = sh;
        }
        __syncthreads();
    }
    // backward BatchNorm: the two means (of g and of g * xhat) into csc / csh; the image-0 workgroup stores dbeta / dgamma
    __device__ __forceinline__ void bn_backward(int e, float* dgamma, float* dbeta) {
        gather(e);
        if (tid < SL) {
            float mg = 0.f, mgx = 0.f;
            if (tid < SLv && !dead) {
                double s1, s2;
                totals(tid, s1, s2);
                const double count = (double)g.N * HW;
                mg = (float)(s1 / count);
                mgx = (float)(s2 / count);
                if (img == 0) { dbeta[c0 + tid] = (float)s1; dgamma[c0 + tid] = (float)s2; }
            }
            csc[tid] = mg; csh[tid] = mgx;
        }
        __syncthreads();
    }
    // q <- relu(q * scale + shift) rounded to bf16 (pixels behind HW: zero)
    __device__ __forceinline__ void bn_relu() {
        float sc[TC][4], sh[TC][4];
#pragma unroll
        for (int i = 0; i < TC; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) { sc[i][r] = csc[co_of(i) + r]; sh[i][r] = csh[co_of(i) + r]; }
#pragma unroll
        for (int i = 0; i < TC; ++i)
#pragma unroll
            for (int j = 0; j < TP; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float v = q[i][j][r] * sc[i][r] + sh[i][r];
                    q[i][j][r] = px_of(j) < HW ? bfround(v > 0.f ? v : 0.f) : 0.f;
                }
    }
    // leave: the last workgroup of the image resets the image's counters (with the same kind of access the exchanges use); the last
    // workgroup of the launch resets the tickets and moves the epoch base past this launch's tags
    __device__ __forceinline__ void leave() {
        if (working) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (tid == 0) {
                const unsigned old = LOCAL ? __hip_atomic_fetch_add(cnt + 96, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)
                                           : __hip_atomic_fetch_add(cnt + 96, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (old == (unsigned)g.NS - 1u) {
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        if (LOCAL) __hip_atomic_exchange(cnt + k * 32, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                        else __hip_atomic_store(cnt + k * 32, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    }
                }
            }
            if (touched == 0x9e3779b9u && g.stamps) g.stamps[255] = touched;      // (keeps the touch loads)
        }
        if (tid == 0) {
            const unsigned old = __hip_atomic_fetch_add(ctl + 32, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (old == gridDim.x - 1u) {
                for (int k = 0; k < 8; ++k) __hip_atomic_store(ctl + 96 + 32 * k, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(ctl + 32, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(ctl, ebase + 3u * XS_MAXB + 8u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
    }
};

// ---------------------------------------------------------------------------------------------------------------------------------
// forward
// ---------------------------------------------------------------------------------------------------------------------------------
template <int SL, int HWP, bool LOCAL>
__global__ __launch_bounds__(XS_THREADS) void xstage_fwd_kernel(const XsArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    typedef XsWg<SL, HWP, LOCAL> WG;
    constexpr int TC = WG::TC, TP = WG::TP, WGP = WG::WGP;
    WG w(p.g, smem);
    if (w.working) {
        const int C = w.C, HW = w.HW, Cs = w.Cs, c0 = w.c0, SLv = w.SLv, img = w.img;
        const long MC = (long)p.g.N * HW * C;
        w.issue_w(p.blk[0].w1);
        for (int b = 0; b < p.nb && !w.dead; ++b) {
            const XsBlock& B = p.blk[b];
            w.cur_b = b;
            w.refresh();
            const int tid = w.tid;
            const bf16* xin = b == 0 ? p.x0 : p.out + (long)(b - 1) * MC;
            float* const coefb = p.coef + (long)b * 12 * C;
            w.stamp(b, 0);
            // conv_block_2's operand fragments, requested at the top of the block (in front of conv_block_3's weight DMAs in the prefetch
            // waves' queues; used behind the first statistics exchange)
            bf16x8 wf[TC][5];
            w.load_wf(B.w2, wf);
            // ---------------------------------------------------------------- phase A: conv_block_1 + BatchNorm + ReLU
            if (b > 0) { w.await(0, (unsigned)(p.g.NS * b)); if (w.dead) break; }
            w.stamp(b, 1);
            w.gemm(((p.g.dbg & 16) ? p.x0 : xin) + (long)img * HW * C);
            w.stamp(b, 2);
            w.acc_to_q(false);
            w.template tile_sums<0>();
            w.stage_tile(false);
            __syncthreads();
            w.publish(3 * b + 0);
            w.issue_w(B.w3);                              // conv_block_3's slice (waves 4-7, behind the publish): lands under phases A and B
            w.store_tile(p.z1 + (long)b * MC, false);
            w.bn_forward(3 * b + 0, B.g1, B.b1, B.rm1, B.rv1, coefb, p.eps, p.momentum);
            if (w.dead) break;
            w.stamp(b, 3);
            // SE weights of the slice (fp32), requested behind the FIRST statistics exchange (they land under the grouped conv; in front of
            // the second exchange's gather they would hold its granule loads back in the in-order queue) and used behind the second.
            // First layer: 16-byte pieces of the [Cs][SL] slab, piece index tid + 512 i -> (row j, piece pc): a wave reads whole
            // 256-byte (128-byte) row slices; the PPJ lanes of a row meet in a lane-group sum.  Second layer: thread (c = tid / PARTS, part)
            // owns the float2 elements part + PARTS i of row c0 + c: 8 (16) neighbouring lanes read 64 (128) contiguous bytes.
            constexpr int PPJ = SL / 4, NL1 = 256 * PPJ / XS_THREADS, PARTS = XS_THREADS / SL, NL2 = 128 / PARTS;
            f32x4 w1r[NL1];
            float w2x[NL2], w2y[NL2];
#pragma unroll
            for (int i = 0; i < NL1; ++i) {
                const int pi = tid + XS_THREADS * i, j = pi / PPJ, pc = pi - j * PPJ;
                w1r[i] = (j < Cs && pc * 4 < SLv) ? *reinterpret_cast<const f32x4*>(B.sw1 + (long)j * C + c0 + pc * 4) : (f32x4){0.f, 0.f, 0.f, 0.f};
            }
            {
                const int c = tid / PARTS, part = tid % PARTS;
#pragma unroll
                for (int i = 0; i < NL2; ++i) {
                    const int f = part + PARTS * i;
                    float2 v = make_float2(0.f, 0.f);
                    if (c < SLv && 2 * f < Cs) v = *reinterpret_cast<const float2*>(B.sw2 + (long)(c0 + c) * Cs + 2 * f);
                    w2x[i] = v.x; w2y[i] = v.y;
                }
            }
            const float sb1v = tid < Cs ? B.sb1[tid] : 0.f;
            const float sb2v = (tid / PARTS) < SLv ? B.sb2[c0 + tid / PARTS] : 0.f;
            // a = relu(bn1(z1)) into the halo tile (zero border = the conv's padding), and out to memory for the weight gradient
            w.zero_halo();
            w.bn_relu();
            __syncthreads();                              // the tile is zero
            w.stage_tile(true);
            __syncthreads();
            w.store_tile(p.a + (long)b * MC, true);
            // ---------------------------------------------------------------- phase B: grouped 3x3 conv + BatchNorm + ReLU + SE
            w.gconv(wf);
            w.stamp(b, 4);
            w.acc_to_q(true);
            w.template tile_sums<0>();
            __syncthreads();                              // every wave is done reading the halo tile
            w.publish(3 * b + 1);
            w.touch_tile(p.bg + (long)b * MC);
            w.stage_tile(false);
            __syncthreads();
            w.store_tile(p.z2 + (long)b * MC, false);
            w.bn_forward(3 * b + 1, B.g2, B.b2, B.rm2, B.rv2, coefb + 4 * C, p.eps, p.momentum);
            if (w.dead) break;
            w.stamp(b, 5);
            w.bn_relu();
            w.template tile_sums<1>();                    // squeeze: per-channel sums of b over the image
            __syncthreads();
            if (tid < SL) {
                float v = 0.f;
#pragma unroll
                for (int k = 0; k < WGP; ++k) v += w.red[(k * 2) * SL + tid];
                v *= p.alpha;
                w.lpool[tid] = tid < SLv ? v : 0.f;
                if (tid < SLv) p.pooled[((long)b * p.g.N + img) * C + c0 + tid] = v;
            }
            __syncthreads();
            {   // first SE layer: this slice's share of every hidden unit
                float* hp = reinterpret_cast<float*>(p.g.ws + XS_HP_OFF) + ((long)img * XS_MAX_SLICES + w.slice) * 256;
#pragma unroll
                for (int i = 0; i < NL1; ++i) {
                    const int pi = tid + XS_THREADS * i, j = pi / PPJ, pc = pi - j * PPJ;
                    const f32x4 pv = *reinterpret_cast<const f32x4*>(w.lpool + pc * 4);
                    float d = w1r[i][0] * pv[0] + w1r[i][1] * pv[1] + w1r[i][2] * pv[2] + w1r[i][3] * pv[3];
#pragma unroll
                    for (int m = 1; m < PPJ; m <<= 1) d += __shfl_xor(d, m);
                    if (pc == 0 && j < Cs) hp[j] = d;
                }
            }
            w.arrive(1);
            w.await(1, (unsigned)(p.g.NS * (b + 1)));
            if (w.dead) break;
            if (tid < 256) {
                float s = 0.f;
                if (tid < Cs) {
                    float part[XS_MAX_SLICES];
#pragma unroll
                    for (int s2 = 0; s2 < XS_MAX_SLICES; ++s2)
                        part[s2] = s2 < p.g.NS ? __hip_atomic_load((gf32*)(p.g.ws + XS_HP_OFF) + ((long)img * XS_MAX_SLICES + s2) * 256 + tid,
                                                                   __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
                                               : 0.f;
#pragma unroll
                    for (int s2 = 0; s2 < XS_MAX_SLICES; ++s2) s += part[s2];
                    s += sb1v;
                    s = s > 0.f ? s : 0.f;
                    if (w.slice == 0) p.hid[((long)b * p.g.N + img) * Cs + tid] = s;
                }
                w.lh[tid] = s;
            }
            __syncthreads();
            {   // second SE layer for the slice's own channels
                const int c = tid / PARTS, part = tid % PARTS;
                float s = 0.f;
#pragma unroll
                for (int i = 0; i < NL2; ++i) {
                    const int f = part + PARTS * i;
                    s += w2x[i] * w.lh[2 * f] + w2y[i] * w.lh[2 * f + 1];
                }
#pragma unroll
                for (int m = 1; m < PARTS; m <<= 1) s += __shfl_xor(s, m);
                if (part == 0) {
                    const float gt = 1.f / (1.f + __expf(-(s + sb2v)));
                    w.lgate[c] = gt;
                    if (c < SLv) p.gate[((long)b * p.g.N + img) * C + c0 + c] = gt;
                }
            }
            __syncthreads();
#pragma unroll
            for (int i = 0; i < TC; ++i)
#pragma unroll
                for (int j = 0; j < TP; ++j)
#pragma unroll
                    for (int r = 0; r < 4; ++r) w.q[i][j][r] = bfround(w.q[i][j][r] * w.lgate[w.co_of(i) + r]);
            w.stage_tile(false);
            __syncthreads();
            w.store_tile(p.bg + (long)b * MC, false, false);
            w.arrive(2);
            w.stamp(b, 6);
            // ---------------------------------------------------------------- phase C: conv_block_3 + BatchNorm + shortcut + ReLU
            bf16x4 xr[TC][TP];                            // the shortcut operand, requested in front of the GEMM
            w.load_tile(xin, xr);
            w.touch_tile(p.out + (long)b * MC);
            w.await(2, (unsigned)(p.g.NS * (b + 1)));
            if (w.dead) break;
            w.stamp(b, 7);
            w.gemm(((p.g.dbg & 16) ? p.x0 : p.bg + (long)b * MC) + (long)img * HW * C);
            w.stamp(b, 8);
            w.acc_to_q(false);
            w.template tile_sums<0>();
            w.stage_tile(false);
            __syncthreads();
            w.publish(3 * b + 2);
            if (b + 1 < p.nb) w.issue_w(p.blk[b + 1].w1); // the next block's first slice (waves 4-7, behind the publish)
            w.store_tile(p.z3 + (long)b * MC, false);
            w.bn_forward(3 * b + 2, B.g3, B.b3, B.rm3, B.rv3, coefb + 8 * C, p.eps, p.momentum);
            if (w.dead) break;
            w.stamp(b, 9);
            {
                float sc[TC][4], sh[TC][4];
#pragma unroll
                for (int i = 0; i < TC; ++i)
#pragma unroll
                    for (int r = 0; r < 4; ++r) { sc[i][r] = w.csc[w.co_of(i) + r]; sh[i][r] = w.csh[w.co_of(i) + r]; }
#pragma unroll
                for (int i = 0; i < TC; ++i)
#pragma unroll
                    for (int j = 0; j < TP; ++j)
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const float v = w.q[i][j][r] * sc[i][r] + sh[i][r] + bf2f(xr[i][j][r]);
                            w.q[i][j][r] = v > 0.f ? v : 0.f;
                        }
            }
            w.stage_tile(false);
            __syncthreads();
            w.store_tile(p.out + (long)b * MC, false, false);
            w.arrive(0, true);                            // (the prefetch waves' DMAs of the next slice stay in flight)
            w.stamp(b, 10);
        }
    }
    w.leave();
}

// ---------------------------------------------------------------------------------------------------------------------------------
// backward: XBlockFn.backward (ops/backbone.py) of the same blocks, last block first.  Per block
//   g = dout [out > 0];  BatchNorm-3 backward (sums of g, g xhat <- all-gather) -> dz3 -> memory (operand of the GEMM below and of dW3)
//   dbg = dz3 W3  (GEMM over conv_block_3's couts: the transposed pack wt3, slice = conv_block_3's INPUT channels)
//   SE backward: dgate = sum_px dbg b, dpre2 = dgate g (1 - g); the slice's share of dhid = W2^T dpre2 <- exchanged inside the image;
//   dpre1 = dhid [hid > 0]; dpool = W1^T dpre1;  g2 = (dbg gate + dpool / HW) [bn2(z2) > 0];  BatchNorm-2 backward -> dz2 (halo tile + memory)
//   da = grouped conv of dz2 with the flipped pack wd2;  g1 = da [bn1(z1) > 0];  BatchNorm-1 backward -> dz1 -> memory
//   dx = dz1 W1 + g  (transposed pack wt1): the next block's dout, same workgroup, kept in registers; the first block's goes to memory
// ---------------------------------------------------------------------------------------------------------------------------------
template <int SL, int HWP, bool LOCAL>
__global__ __launch_bounds__(XS_THREADS) void xstage_bwd_kernel(const XbArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    typedef XsWg<SL, HWP, LOCAL> WG;
    constexpr int TC = WG::TC, TP = WG::TP, WGP = WG::WGP;
    WG w(p.g, smem);
    if (w.working) {
        const int C = w.C, HW = w.HW, Cs = w.Cs, c0 = w.c0, SLv = w.SLv, img = w.img;
        const long MC = (long)p.g.N * HW * C;
        const float inv_hw = 1.0f / (float)HW;
        // Register diet: tiles stay bf16 (one register per two values); xhat and the ReLU masks are recomputed from them where they are used.
        bf16x4 dqb[TC][TP];                               // dout of the block in hand: the previous block's dx (the launch's dout for the last block)
        w.load_tile(p.dout, dqb);
        w.issue_w(p.blk[p.nb - 1].wt3);
        // the block's output and conv_block_3 output tiles (BatchNorm-3 backward): cold in HBM, requested one GEMM ahead
        bf16x4 ty[TC][TP], tz[TC][TP];
        w.load_tile(p.out + (long)(p.nb - 1) * MC, ty);
        w.load_tile(p.z3 + (long)(p.nb - 1) * MC, tz);
        // the forward coefficients (scale, shift, mean, rstd) of BatchNorm `k` of block `b` for the slice's channels -> lcoef
        auto load_coef = [&](int b, int k) {
            if (w.tid < 4 * SL) {
                const int r = w.tid / SL, c = w.tid - r * SL;
                w.lcoef[r * SL + c] = c < SLv ? p.coef[((long)b * 3 + k) * 4 * C + (long)r * C + c0 + c] : 0.f;
            }
        };
        // q <- scale (q - mean_g - xhat mean_gx) with xhat from the bf16 tile t (BatchNorm backward apply)
        auto bn_apply = [&](const bf16x4 (&t)[TC][TP]) {
#pragma unroll
            for (int i = 0; i < TC; ++i)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int co = w.co_of(i) + r;
                    const float sc = w.lcoef[co], mu = w.lcoef[2 * SL + co], rs = w.lcoef[3 * SL + co], mg = w.csc[co], mgx = w.csh[co];
#pragma unroll
                    for (int j = 0; j < TP; ++j) {
                        const float xh = (bf2f(t[i][j][r]) - mu) * rs;
                        w.q[i][j][r] = w.px_of(j) < HW ? sc * (w.q[i][j][r] - mg - xh * mgx) : 0.f;
                    }
                }
        };
        // red <- (sum q, sum q xhat) over the workgroup's pixels, xhat from the bf16 tile t
        auto bn_sums = [&](const bf16x4 (&t)[TC][TP]) {
#pragma unroll
            for (int i = 0; i < TC; ++i)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int co = w.co_of(i) + r;
                    const float mu = w.lcoef[2 * SL + co], rs = w.lcoef[3 * SL + co];
                    float s1 = 0.f, s2 = 0.f;
#pragma unroll
                    for (int j = 0; j < TP; ++j) { s1 += w.q[i][j][r]; s2 += w.q[i][j][r] * ((bf2f(t[i][j][r]) - mu) * rs); }
                    s1 = row16_sum(s1);
                    s2 = row16_sum(s2);
                    if ((w.lane & 15) == 0) { w.red[(w.wp * 2 + 0) * SL + co] = s1; w.red[(w.wp * 2 + 1) * SL + co] = s2; }
                }
        };
        for (int it = 0; it < p.nb && !w.dead; ++it) {
            const int b = p.nb - 1 - it;
            const XbBlock& B = p.blk[b];
            w.cur_b = b;
            w.refresh();
            const int tid = w.tid;
            float* const dgb = p.dgb + (long)b * 6 * C;
            w.stamp(it, 0);
            // ---------------------------------------------------------------- BatchNorm-3 backward (mask: the block's output)
            load_coef(b, 2);
            w.touch_tile(p.dz3 + (long)b * MC);
            bf16x4 gres[TC][TP];                          // g = dout [out > 0]: the gradient of the identity branch, added to dx at the end
#pragma unroll
            for (int i = 0; i < TC; ++i)
#pragma unroll
                for (int j = 0; j < TP; ++j)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float gv = bf2f(ty[i][j][r]) > 0.f ? bf2f(dqb[i][j][r]) : 0.f;
                        w.q[i][j][r] = gv;
                        gres[i][j][r] = f2bf(gv);
                    }
            __syncthreads();                              // lcoef
            bn_sums(tz);
            __syncthreads();
            w.publish(3 * it + 0);
            w.bn_backward(3 * it + 0, dgb + 4 * C, dgb + 5 * C);
            if (w.dead) break;
            w.stamp(it, 1);
            bn_apply(tz);
            w.stage_tile(false);
            __syncthreads();
            w.store_tile(p.dz3 + (long)b * MC, false, false);
            w.arrive(0);
            // ---------------------------------------------------------------- dbg = dz3 W3, SE backward, BatchNorm-2 backward
            // SE weights of the slice, the z2 tile and the gate / hidden vectors: requested in front of the first GEMM (they land under it; in
            // front of a gather they held its granule loads back for the whole trip -- 29 MB of SE weights leave HBM at once):
            //   dhid share: thread (j = tid & 255, half) owns W2[c0 + half SL/2 + k][j]; dpool: thread (c = tid % SL, part) owns W1[part JP + k][c0 + c]
            constexpr int HC = SL / 2, PARTS = XS_THREADS / SL, JP = 256 / PARTS;
            float w2r[HC], w1r[JP];
            auto load_se = [&]() {
                const int j = tid & 255, hc0 = (tid >> 8) * HC;
#pragma unroll
                for (int k = 0; k < HC; ++k) w2r[k] = (j < Cs && hc0 + k < SLv) ? B.sw2[(long)(c0 + hc0 + k) * Cs + j] : 0.f;
                const int c = tid % SL, j0 = (tid / SL) * JP;
#pragma unroll
                for (int k = 0; k < JP; ++k) w1r[k] = (c < SLv && j0 + k < Cs) ? B.sw1[(long)(j0 + k) * C + c0 + c] : 0.f;
            };
            load_se();
            const float gatev = tid < SLv ? p.gate[((long)b * p.g.N + img) * C + c0 + tid] : 0.f;
            const float hidv = (tid < 256 && tid < Cs) ? p.hid[((long)b * p.g.N + img) * Cs + tid] : 0.f;
            bf16x4 tz2[TC][TP];
            w.load_tile(p.z2 + (long)b * MC, tz2);
            w.await(0, (unsigned)(p.g.NS * (it + 1)));
            if (w.dead) break;
            w.stamp(it, 2);
            w.gemm(p.dz3 + (long)b * MC + (long)img * HW * C);
            w.stamp(it, 3);
            w.issue_w(B.wt1);                             // conv_block_1's transposed slice (waves 4-7): lands under the SE / grouped-conv phases
            w.acc_to_q(true);                             // dbg (bf16 as the chain stores it)
            load_coef(b, 1);
            if (tid < SL) w.lgate[tid] = gatev;
            __syncthreads();
            {   // red[.][1] = sum_px dbg * b, b = relu(bn2(z2)) as the forward saw it  (the gate gradient)
#pragma unroll
                for (int i = 0; i < TC; ++i)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int co = w.co_of(i) + r;
                        const float sc = w.lcoef[co], sh = w.lcoef[SL + co];
                        float s2 = 0.f;
#pragma unroll
                        for (int j = 0; j < TP; ++j) {
                            const float pre = bf2f(tz2[i][j][r]) * sc + sh;
                            s2 += w.q[i][j][r] * bfround(pre > 0.f ? pre : 0.f);
                        }
                        s2 = row16_sum(s2);
                        if ((w.lane & 15) == 0) w.red[(w.wp * 2 + 1) * SL + co] = s2;
                    }
            }
            __syncthreads();
            if (tid < SL) {
                float v = 0.f;
#pragma unroll
                for (int k = 0; k < WGP; ++k) v += w.red[(k * 2 + 1) * SL + tid];
                const float gt = w.lgate[tid];
                v *= gt * (1.f - gt);
                w.lpool[tid] = tid < SLv ? v : 0.f;                                  // dpre2 of the slice's channels
                if (tid < SLv) p.dpre2[((long)b * p.g.N + img) * C + c0 + tid] = v;
            }
            __syncthreads();
            {   // this slice's share of dhid = W2^T dpre2
                const int j = tid & 255, half = tid >> 8;
                float s = 0.f;
#pragma unroll
                for (int k = 0; k < HC; ++k) s += w2r[k] * w.lpool[half * HC + k];
                w.lvec[half * 256 + j] = s;
            }
            __syncthreads();
            if (tid < 256 && tid < Cs)
                reinterpret_cast<float*>(p.g.ws + XS_HP_OFF)[((long)img * XS_MAX_SLICES + w.slice) * 256 + tid] = w.lvec[tid] + w.lvec[256 + tid];
            w.arrive(1, true);                            // (stored by waves 0-3; waves 4-7 have conv_block_1's slice in flight)
            w.await(1, (unsigned)(p.g.NS * (it + 1)));
            if (w.dead) break;
            if (tid < 256) {
                float s = 0.f;
                if (tid < Cs) {
                    float part[XS_MAX_SLICES];
#pragma unroll
                    for (int s2 = 0; s2 < XS_MAX_SLICES; ++s2)
                        part[s2] = s2 < p.g.NS ? __hip_atomic_load((gf32*)(p.g.ws + XS_HP_OFF) + ((long)img * XS_MAX_SLICES + s2) * 256 + tid,
                                                                   __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
                                               : 0.f;
#pragma unroll
                    for (int s2 = 0; s2 < XS_MAX_SLICES; ++s2) s += part[s2];
                    if (!(hidv > 0.f)) s = 0.f;
                    if (w.slice == 0) p.dpre1[((long)b * p.g.N + img) * Cs + tid] = s;
                }
                w.lh[tid] = s;
            }
            __syncthreads();
            {   // dpool[c] = sum_j W1[j][c] dpre1[j]: the parts meet in LDS
                const int c = tid % SL, part = tid / SL;
                float s = 0.f;
#pragma unroll
                for (int k = 0; k < JP; ++k) s += w1r[k] * w.lh[part * JP + k];
                w.lvec[part * SL + c] = s;                // [PARTS][SL] = 512 floats
            }
            __syncthreads();
            if (tid < SL) {
                float s = 0.f;
#pragma unroll
                for (int k = 0; k < PARTS; ++k) s += w.lvec[k * SL + tid];
                w.lpool[tid] = s * inv_hw;                // dpool / HW
            }
            __syncthreads();
#pragma unroll
            for (int i = 0; i < TC; ++i)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int co = w.co_of(i) + r;
                    const float gt = w.lgate[co], dp = w.lpool[co], sc = w.lcoef[co], sh = w.lcoef[SL + co];
#pragma unroll
                    for (int j = 0; j < TP; ++j) {
                        const float gv = bfround(w.q[i][j][r] * gt + dp);
                        w.q[i][j][r] = (bf2f(tz2[i][j][r]) * sc + sh > 0.f && w.px_of(j) < HW) ? gv : 0.f;
                    }
                }
            bn_sums(tz2);
            __syncthreads();
            w.publish(3 * it + 1);
            bf16x4 tz1[TC][TP];                           // requested behind the publish (16 KB per workgroup), used behind the grouped conv
            w.load_tile(p.z1 + (long)b * MC, tz1);
            bf16x8 wf[TC][5];                             // the grouped conv's flipped operand (wide tiles: behind the gather -- 40 registers)
            if (TP <= 2) w.load_wf(B.wd2, wf);
            w.bn_backward(3 * it + 1, dgb + 2 * C, dgb + 3 * C);
            if (w.dead) break;
            if (TP > 2) w.load_wf(B.wd2, wf);
            w.stamp(it, 4);
            w.zero_halo();
            bn_apply(tz2);
            __syncthreads();                              // the tile is zero; lcoef readers are done
            w.stage_tile(true);
            load_coef(b, 0);
            __syncthreads();
            w.store_tile(p.dz2 + (long)b * MC, true);
            // ---------------------------------------------------------------- da = dgrad of the grouped conv, BatchNorm-1 backward
            w.gconv(wf);
            w.stamp(it, 5);
            w.acc_to_q(true);                             // da
            w.touch_tile(p.dz1 + (long)b * MC);
#pragma unroll
            for (int i = 0; i < TC; ++i)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int co = w.co_of(i) + r;
                    const float sc = w.lcoef[co], sh = w.lcoef[SL + co];
#pragma unroll
                    for (int j = 0; j < TP; ++j)
                        if (!(sc * bf2f(tz1[i][j][r]) + sh > 0.f)) w.q[i][j][r] = 0.f;
                }
            bn_sums(tz1);
            __syncthreads();                              // (also: every wave is done reading the halo tile)
            w.publish(3 * it + 2);
            w.bn_backward(3 * it + 2, dgb, dgb + C);
            if (w.dead) break;
            w.stamp(it, 6);
            bn_apply(tz1);
            w.stage_tile(false);
            __syncthreads();
            w.store_tile(p.dz1 + (long)b * MC, false, false);
            w.arrive(2);
            // ---------------------------------------------------------------- dx = dz1 W1 + g
            if (b > 0) {                                  // the next block's BatchNorm-3 operands: land under the GEMM
                w.load_tile(p.out + (long)(b - 1) * MC, ty);
                w.load_tile(p.z3 + (long)(b - 1) * MC, tz);
            }
            w.await(2, (unsigned)(p.g.NS * (it + 1)));
            if (w.dead) break;
            w.stamp(it, 7);
            w.gemm(p.dz1 + (long)b * MC + (long)img * HW * C);
            w.stamp(it, 8);
            if (b > 0) w.issue_w(p.blk[b - 1].wt3);       // the next block's first slice (waves 4-7)
#pragma unroll
            for (int i = 0; i < TC; ++i)
#pragma unroll
                for (int j = 0; j < TP; ++j)
#pragma unroll
                    for (int r = 0; r < 4; ++r) dqb[i][j][r] = f2bf(bfround(w.acc[i][j][r]) + bf2f(gres[i][j][r]));
            if (b == 0) {
#pragma unroll
                for (int i = 0; i < TC; ++i)
#pragma unroll
                    for (int j = 0; j < TP; ++j)
#pragma unroll
                        for (int r = 0; r < 4; ++r) w.q[i][j][r] = bf2f(dqb[i][j][r]);
                w.stage_tile(false);
                __syncthreads();
                w.store_tile(p.dx, false, false);
            }
            w.stamp(it, 9);
        }
    }
    w.leave();
}

// ---------------------------------------------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------------------------------------------
static int xs_variant(int N, int H, int W, int C, int Cs) {
    const int KP = (C + 31) / 32 * 32, HW = H * W;
    if (N < 8 || N > XS_MAX_IMG || (N & 7) || (C & 7) || Cs < 1 || Cs > 256 || (Cs & 1) || H < 1 || W < 1) return 0;
    const int ipx = N / 8;
    if (HW <= 128 && N <= 16 && 64 * ((KP + 63) / 64 * 64) * 2 + 2 * 128 * 128 + 8192 <= 160 * 1024 && ipx * ((C + 63) / 64) <= 32 &&
        (C + 63) / 64 <= XS_MAX_SLICES && (H + 2) * (W + 2) * 144 <= 2 * 128 * 128 && 128 * 144 + N * 128 * 4 <= 2 * 128 * 128)
        return 1;
    if (HW <= 512 && 32 * ((KP + 63) / 64 * 64) * 2 + 2 * 512 * 128 + 8192 <= 160 * 1024 && ipx * ((C + 31) / 32) <= 32 &&
        (C + 31) / 32 <= XS_MAX_SLICES && (H + 2) * (W + 2) * 80 <= 2 * 512 * 128)
        return 2;
    return 0;
}
extern "C" long hn_xstage_ws_bytes(void) { return XS_WS_BYTES; }
extern "C" int hn_xstage_supported(int N, int H, int W, int C, int Cs) { return xs_variant(N, H, W, C, Cs); }

static std::atomic<unsigned long long> g_xs_optin{0};
static bool xs_optin() {
    return lds_optin(g_xs_optin, {(const void*)xstage_fwd_kernel<64, 128, true>, (const void*)xstage_fwd_kernel<32, 512, true>,
                                  (const void*)xstage_fwd_kernel<64, 128, false>, (const void*)xstage_fwd_kernel<32, 512, false>,
                                  (const void*)xstage_bwd_kernel<64, 128, true>, (const void*)xstage_bwd_kernel<32, 512, true>,
                                  (const void*)xstage_bwd_kernel<64, 128, false>, (const void*)xstage_bwd_kernel<32, 512, false>});
}
static void xs_common(XsCommon& g, int var, int N, int H, int W, int C, int Cs, void* ws, long* stamps, int& mode) {
    g.N = N; g.H = H; g.W = W; g.C = C; g.KP = (C + 31) / 32 * 32; g.Cs = Cs;
    g.ws = (char*)ws;
    g.ipx = N / 8;
    g.NS = var == 1 ? (C + 63) / 64 : (C + 31) / 32;
    g.stamps = (unsigned long long*)stamps;
    g.dbg = mode & ~1;
    mode &= 1;
}
static size_t xs_lds(int var, int KP) {
    return (size_t)(var == 1 ? 64 : 32) * ((KP + 63) / 64 * 64) * 2 + 2 * (size_t)(var == 1 ? 128 : 512) * 128 + 8192;
}

// tab: HOST table nb x 19 int64 = the members of XsBlock in declaration order.  mode 0: XCD-local counters (the product form), 1: agent-scope
// counters with release / acquire fences (placement independent).
extern "C" int hn_xstage_fwd(const long* tab, int nb, const void* x0, void* z1, void* a, void* z2, void* bg, void* z3, void* out, float* coef,
                             float* pooled, float* hid, float* gate, int N, int H, int W, int C, int Cs, float eps, float momentum, float alpha,
                             void* ws, long* stamps, int mode, hipStream_t st) {
    HN_CHECK_ARG(tab && nb >= 1 && nb <= XS_MAXB && x0 && z1 && a && z2 && bg && z3 && out && coef && pooled && hid && gate && ws);
    const int var = xs_variant(N, H, W, C, Cs);
    if (!var) return HN_ERR_UNSUPPORTED;
    XsArgs p;
    for (int b = 0; b < nb; ++b) {
        const long* t = tab + 19 * b;
        for (int k = 0; k < 19; ++k) HN_CHECK_ARG(t[k] != 0);
        XsBlock& B = p.blk[b];
        B.w1 = (const bf16*)t[0]; B.w2 = (const bf16*)t[1]; B.w3 = (const bf16*)t[2];
        B.sw1 = (const float*)t[3]; B.sb1 = (const float*)t[4]; B.sw2 = (const float*)t[5]; B.sb2 = (const float*)t[6];
        B.g1 = (const float*)t[7]; B.b1 = (const float*)t[8]; B.rm1 = (float*)t[9]; B.rv1 = (float*)t[10];
        B.g2 = (const float*)t[11]; B.b2 = (const float*)t[12]; B.rm2 = (float*)t[13]; B.rv2 = (float*)t[14];
        B.g3 = (const float*)t[15]; B.b3 = (const float*)t[16]; B.rm3 = (float*)t[17]; B.rv3 = (float*)t[18];
    }
    p.nb = nb;
    xs_common(p.g, var, N, H, W, C, Cs, ws, stamps, mode);
    p.x0 = (const bf16*)x0;
    p.z1 = (bf16*)z1; p.a = (bf16*)a; p.z2 = (bf16*)z2; p.bg = (bf16*)bg; p.z3 = (bf16*)z3; p.out = (bf16*)out;
    p.coef = coef; p.pooled = pooled; p.hid = hid; p.gate = gate;
    p.eps = eps; p.momentum = momentum; p.alpha = alpha;
    if (!xs_optin()) return HN_ERR_LAUNCH;
    const size_t lds = xs_lds(var, p.g.KP);
    if (var == 1 && mode == 0) hipLaunchKernelGGL((xstage_fwd_kernel<64, 128, true>), dim3(XS_GRID), dim3(XS_THREADS), lds, st, p);
    else if (var == 1) hipLaunchKernelGGL((xstage_fwd_kernel<64, 128, false>), dim3(XS_GRID), dim3(XS_THREADS), lds, st, p);
    else if (mode == 0) hipLaunchKernelGGL((xstage_fwd_kernel<32, 512, true>), dim3(XS_GRID), dim3(XS_THREADS), lds, st, p);
    else hipLaunchKernelGGL((xstage_fwd_kernel<32, 512, false>), dim3(XS_GRID), dim3(XS_THREADS), lds, st, p);
    HN_LAUNCH_CHECK();
}

// tab: HOST table nb x 5 int64 = {wt1 (transposed pack of conv_block_1), wd2 (hn_gconv_pack_diag's data-gradient operand), wt3, se.1.weight,
// se.3.weight} per block, in FORWARD block order; the launch walks the blocks last to first.
extern "C" int hn_xstage_bwd(const long* tab, int nb, const void* dout, const void* z1, const void* z2, const void* z3, const void* out,
                             const float* coef, const float* hid, const float* gate, void* dz1, void* dz2, void* dz3, void* dx, float* dgb,
                             float* dpre2, float* dpre1, int N, int H, int W, int C, int Cs, void* ws, long* stamps, int mode, hipStream_t st) {
    HN_CHECK_ARG(tab && nb >= 1 && nb <= XS_MAXB && dout && z1 && z2 && z3 && out && coef && hid && gate && dz1 && dz2 && dz3 && dx && dgb &&
                 dpre2 && dpre1 && ws);
    const int var = xs_variant(N, H, W, C, Cs);
    if (!var) return HN_ERR_UNSUPPORTED;
    XbArgs p;
    for (int b = 0; b < nb; ++b) {
        const long* t = tab + 5 * b;
        for (int k = 0; k < 5; ++k) HN_CHECK_ARG(t[k] != 0);
        XbBlock& B = p.blk[b];
        B.wt1 = (const bf16*)t[0]; B.wd2 = (const bf16*)t[1]; B.wt3 = (const bf16*)t[2]; B.sw1 = (const float*)t[3]; B.sw2 = (const float*)t[4];
    }
    p.nb = nb;
    xs_common(p.g, var, N, H, W, C, Cs, ws, stamps, mode);
    p.dout = (const bf16*)dout;
    p.z1 = (const bf16*)z1; p.z2 = (const bf16*)z2; p.z3 = (const bf16*)z3; p.out = (const bf16*)out;
    p.coef = coef; p.hid = hid; p.gate = gate;
    p.dz1 = (bf16*)dz1; p.dz2 = (bf16*)dz2; p.dz3 = (bf16*)dz3; p.dx = (bf16*)dx;
    p.dgb = dgb; p.dpre2 = dpre2; p.dpre1 = dpre1;
    if (!xs_optin()) return HN_ERR_LAUNCH;
    const size_t lds = xs_lds(var, p.g.KP);
    if (var == 1 && mode == 0) hipLaunchKernelGGL((xstage_bwd_kernel<64, 128, true>), dim3(XS_GRID), dim3(XS_THREADS), lds, st, p);
    else if (var == 1) hipLaunchKernelGGL((xstage_bwd_kernel<64, 128, false>), dim3(XS_GRID), dim3(XS_THREADS), lds, st, p);
    else if (mode == 0) hipLaunchKernelGGL((xstage_bwd_kernel<32, 512, true>), dim3(XS_GRID), dim3(XS_THREADS), lds, st, p);
    else hipLaunchKernelGGL((xstage_bwd_kernel<32, 512, false>), dim3(XS_GRID), dim3(XS_THREADS), lds, st, p);
    HN_LAUNCH_CHECK();
}
