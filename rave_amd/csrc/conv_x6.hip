// Host side of the bf16x6 convolution path (kernels: conv_x6_kernel.inc): tile / split-K plan and launch.
#include <atomic>
#include <cstdlib>
#include <mutex>
#include "conv_params.hpp"
#include "conv2d_x6.hpp"

// one translation unit per (input stride, tile shape): conv_x6_i<IS>_<TM><TN><WM>.hip
int rh_splitk_finalize_launch(ConvP& p, hipStream_t stream);
bool rh_x6_launch_i1_121(const ConvP& q, int epi, dim3 grid, size_t lds, hipStream_t stream);
bool rh_x6_launch_i1_221(const ConvP& q, int epi, dim3 grid, size_t lds, hipStream_t stream);
bool rh_x6_launch_i1_321(const ConvP& q, int epi, dim3 grid, size_t lds, hipStream_t stream);
bool rh_x6_launch_i1_122(const ConvP& q, int epi, dim3 grid, size_t lds, hipStream_t stream);
bool rh_x6_launch_i1_222(const ConvP& q, int epi, dim3 grid, size_t lds, hipStream_t stream);
bool rh_x6_launch_i1_322(const ConvP& q, int epi, dim3 grid, size_t lds, hipStream_t stream);
bool rh_x6_launch_i1_211(const ConvP& q, int epi, dim3 grid, size_t lds, hipStream_t stream);
bool rh_x6_launch_i1_311(const ConvP& q, int epi, dim3 grid, size_t lds, hipStream_t stream);
bool rh_x6_launch_i1_212(const ConvP& q, int epi, dim3 grid, size_t lds, hipStream_t stream);
bool rh_x6_launch_i1_312(const ConvP& q, int epi, dim3 grid, size_t lds, hipStream_t stream);
bool rh_x6_launch_i2_121(const ConvP& q, int epi, dim3 grid, size_t lds, hipStream_t stream);
bool rh_x6_launch_i2_221(const ConvP& q, int epi, dim3 grid, size_t lds, hipStream_t stream);
bool rh_x6_launch_i2_321(const ConvP& q, int epi, dim3 grid, size_t lds, hipStream_t stream);
bool rh_x6_launch_i2_122(const ConvP& q, int epi, dim3 grid, size_t lds, hipStream_t stream);
bool rh_x6_launch_i2_222(const ConvP& q, int epi, dim3 grid, size_t lds, hipStream_t stream);
bool rh_x6_launch_i2_322(const ConvP& q, int epi, dim3 grid, size_t lds, hipStream_t stream);
bool rh_x6_launch_i2_211(const ConvP& q, int epi, dim3 grid, size_t lds, hipStream_t stream);
bool rh_x6_launch_i2_311(const ConvP& q, int epi, dim3 grid, size_t lds, hipStream_t stream);
bool rh_x6_launch_i2_212(const ConvP& q, int epi, dim3 grid, size_t lds, hipStream_t stream);
bool rh_x6_launch_i2_312(const ConvP& q, int epi, dim3 grid, size_t lds, hipStream_t stream);
bool rh_x6_launch_i4_121(const ConvP& q, int epi, dim3 grid, size_t lds, hipStream_t stream);
bool rh_x6_launch_i4_221(const ConvP& q, int epi, dim3 grid, size_t lds, hipStream_t stream);
bool rh_x6_launch_i4_321(const ConvP& q, int epi, dim3 grid, size_t lds, hipStream_t stream);
bool rh_x6_launch_i4_122(const ConvP& q, int epi, dim3 grid, size_t lds, hipStream_t stream);
bool rh_x6_launch_i4_222(const ConvP& q, int epi, dim3 grid, size_t lds, hipStream_t stream);
bool rh_x6_launch_i4_322(const ConvP& q, int epi, dim3 grid, size_t lds, hipStream_t stream);
bool rh_x6_launch_i4_211(const ConvP& q, int epi, dim3 grid, size_t lds, hipStream_t stream);
bool rh_x6_launch_i4_311(const ConvP& q, int epi, dim3 grid, size_t lds, hipStream_t stream);
bool rh_x6_launch_i4_212(const ConvP& q, int epi, dim3 grid, size_t lds, hipStream_t stream);
bool rh_x6_launch_i4_312(const ConvP& q, int epi, dim3 grid, size_t lds, hipStream_t stream);

namespace {

struct X6Plan {
    size_t lds;
    int64_t part_bytes;
    int tm, tn, wm, wn, ksplit, chunks_per_split, col_tiles, row_tiles;
};

typedef bool (*X6Launch)(const ConvP&, int, dim3, size_t, hipStream_t);
X6Launch x6_launcher(int is, int tm, int tn, int wm) {
#define RH_X6_S(IS_, TM_, TN_, WM_) if (is == IS_ && tm == TM_ && tn == TN_ && wm == WM_) return rh_x6_launch_i##IS_##_##TM_##TN_##WM_;
#define RH_X6_SHAPES(IS_) \
    RH_X6_S(IS_, 1, 2, 1) RH_X6_S(IS_, 2, 2, 1) RH_X6_S(IS_, 3, 2, 1) RH_X6_S(IS_, 1, 2, 2) RH_X6_S(IS_, 2, 2, 2) RH_X6_S(IS_, 3, 2, 2) \
    RH_X6_S(IS_, 2, 1, 1) RH_X6_S(IS_, 3, 1, 1) RH_X6_S(IS_, 2, 1, 2) RH_X6_S(IS_, 3, 1, 2)
    RH_X6_SHAPES(1) RH_X6_SHAPES(2) RH_X6_SHAPES(4)
#undef RH_X6_SHAPES
#undef RH_X6_S
    return nullptr;
}
// Arrival counters of the K-split launches (x6_combine): one word per output tile, zero whenever no launch that uses it is in
// flight (the last arriver of a tile resets it).  A launch takes the next segment of ONE pool allocated (and zeroed) on first use
// PER DEVICE -- planning calls come first, so never under stream capture -- : launches that may overlap (two streams, graph branches) get
// different segments, and a segment comes round again only after kTicketPool / tiles (thousands of) later launches.  The address
// is baked into a captured graph node; replays of a node serialize.
constexpr unsigned kTicketPool = 1u << 20;
constexpr int kTicketDevices = 64;
struct TicketPool { unsigned* words = nullptr; std::atomic<unsigned> next{0}; std::once_flag once; };
TicketPool* ticket_pool() {                             // the pool of the CURRENT device (one process may drive several)
    static TicketPool pools[kTicketDevices];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kTicketDevices) return nullptr;
    TicketPool* tp = &pools[dev];
    std::call_once(tp->once, [tp] {
        unsigned* q = nullptr;
        if (hipMalloc(&q, kTicketPool * sizeof(unsigned)) == hipSuccess && hipMemset(q, 0, kTicketPool * sizeof(unsigned)) == hipSuccess)
            tp->words = q;
    });
    return tp->words ? tp : nullptr;
}
unsigned* ticket_segment(unsigned n) {
    TicketPool* tp = ticket_pool();
    if (!tp || n > kTicketPool) return nullptr;
    for (;;) {
        unsigned at = tp->next.load(std::memory_order_relaxed);
        const unsigned start = at + n > kTicketPool ? 0u : at;
        if (tp->next.compare_exchange_weak(at, start + n, std::memory_order_relaxed)) return tp->words + start;
    }
}

// K splits up to this many slices are combined inside the launch; finer ones by the finalize launch (RH_X6_COMBINE, read per call)
int combine_max() {
    const char* e = getenv("RH_X6_COMBINE");
    return e ? atoi(e) : 8;
}

int epi_mode(const ConvP& p) { return (p.bias ? 1 : 0) | (p.mul_src ? 2 : 0) | (p.add ? 4 : 0) | (p.out_act == RH_ACT_LEAKY ? 8 : 0); }

bool x6_enabled() {
    const char* e = getenv("RH_CONV_X6");      // read per call: the parity tests flip it at run time
    return !(e && atoi(e) == 0);
}

bool plan_x6(ConvP& p, X6Plan* pl) {
    if (!x6_enabled() || p.x6_mode == 0) return false;
    if (RH_X6_F16 && !p.in_range) return false;       // no range slot for the input (rh_x6_set_ranges): f32-input MFMA kernels
    p.vs = 0;
    p.Mr = p.M;
    // Virtual rows (conv_host.hip: vplan_of; second section of the packed operand): the s output phases become s * M
    // rows of one stride-1 gather with taps in[n + b0 - u], a column's s outputs are contiguous.  The kernel sees a
    // plain mode-1 problem plus `vs`.  When the two phase groups have different bases the run of a column straddles
    // two positions and the row needs ONE MORE column: taken unless that inflates the column tiles (rows of 2^k
    // positions: 257 columns = 3 tiles of 128) -- then the per-phase section runs as before.
    if (p.v_s > 1 && p.x6_mode == 1 && p.nphase == p.v_s && p.is == 1 && p.inner == 1) {
        const int s = p.v_s;
        const int extra = p.v_o0 ? 1 : 0;
        auto tiles = [](int n) {
            if (n >= 128) return (double)rh_cdiv(n, 128);
            int w = 32;
            while (w < n) w <<= 1;
            return w / 128.0;
        };
        const int mvp = (p.Mr * s + 31) & ~31;
        const int tmv = mvp % 96 == 0 ? 3 : (mvp % 64 == 0 ? 2 : 1);
        const bool epi_ok = rh_x6_epi_instantiated(1, tmv, p.in_act == RH_ACT_LEAKY, epi_mode(p) | (s == 2 ? 16 : 32));
        if (epi_ok && tiles(p.ncols + extra) <= 1.07 * tiles(p.ncols)) {
            p.vs = s;
            p.M = p.Mr * s;
            p.Mp = (p.M + 31) & ~31;
            p.ncols += extra;
            p.nphase = 1;
            p.ph_ntaps[0] = p.v_U; p.ph_tap0[0] = 0; p.ph_oph[0] = 0; p.ph_q2ofs[0] = p.v_q2ofs;
            p.ph_maxoff[0] = p.v_b0; p.ph_minoff[0] = p.v_b0 - p.v_U + 1;
            for (int u = 0; u < p.v_U; ++u) p.off[u] = p.v_b0 - u;
        }
    }
    if (p.x6_mode != p.is) return false;
    if (p.in_act == RH_ACT_SNAKE || p.epi_act == RH_ACT_SNAKE) return false;
    if (p.in_act == RH_ACT_LEAKY && !(p.in_slope >= 0.f && p.in_slope <= 1.f)) return false;      // applied as max(x, slope x)
    // epilogue operand combinations there is a kernel instance for (bit 0 bias, 1 derivative, 2 add, 3 output activation):
    // everything the modules use -- anything else takes the f32 kernels
    if (!rh_x6_epi_instantiated(p.is, 3, p.in_act == RH_ACT_LEAKY, epi_mode(p))) return false;
    if (p.is != 1 && (p.inner != 1 || p.nphase != 1)) return false;
    if (((uintptr_t)p.wq & 15) || ((uintptr_t)p.in & 3)) return false;
    const int is = p.is;
    pl->tm = p.Mp % 96 == 0 ? 3 : (p.Mp % 64 == 0 ? 2 : 1);
    int span = 0;
    if (is == 1) {
        for (int i = 0; i < p.nphase; ++i) span = span > p.ph_maxoff[i] - p.ph_minoff[i] ? span : p.ph_maxoff[i] - p.ph_minoff[i];
        span *= p.inner;
    } else {
        span = p.x6_nu - 1;
    }
    // Workgroup tile = (32*tm*wm) rows x (32*tn*wn) columns, wm*wn = 4 waves, wave tile 32*tm x 32*tn.  Two waves of
    // rows when the layer has them; the 64-column wave tile (half the A-fragment reads per MFMA) unless that leaves
    // fewer than ~1.5 workgroups per CU -- then the 32-column one doubles the number of tiles, which beats splitting K
    // (partial sums written and re-read + a finalize launch; measured on the C = 192 ... 768 layers).
    auto shape = [&](int tn, int wm) -> int {        // fills the tile fields of p; returns the number of tiles (0 = does not fit)
        pl->tn = tn; pl->wm = wm; pl->wn = 4 / wm;
        const int BM = 32 * pl->tm * wm, BN = 32 * tn * pl->wn;
        int bnl = BN;
        if (p.ncols < BN) {
            bnl = 32;
            while (bnl < p.ncols) bnl <<= 1;
        }
        p.bnl = bnl;
        p.bnl_shift = __builtin_ctz(bnl);
        p.nb = BN / bnl;
        p.tiles_per_b = rh_cdiv(p.ncols, bnl);
        p.pitch = bnl + span;
        p.x6_P = p.nb * p.pitch;
        const int nq = pl->wn * tn == 8 ? 3 : 2;
        if (2 * p.x6_P > 256 * nq) return 0;
        p.x6_a_units = 2 * kX6P * BM;
        p.x6_b_units = 2 * kX6P * p.x6_P;
        pl->lds = (size_t)(2 * p.x6_a_units + 2 * p.x6_b_units) * 16;
        if (pl->lds > 160 * 1024) return 0;
        pl->col_tiles = rh_cdiv(p.B, p.nb) * p.tiles_per_b;
        pl->row_tiles = rh_cdiv(p.Mp, BM);
        return pl->col_tiles * pl->row_tiles * p.nphase;
    };
    const int wm0 = p.Mp >= 64 * pl->tm ? 2 : 1;
    static const int tn_env = [] { const char* e = getenv("RH_X6_TN"); return e ? atoi(e) : 0; }();
    int blocks = shape(2, wm0);
    // ... for the short reductions only (pointwise and k = 3 convs at few channels: <= 64 steps): with a long K loop
    // per tile the split's fixed cost is small and the larger wave tile wins (measured per layer, profiles/)
    const int total_steps = ((p.C * is) >> 4) * (is == 1 ? p.ph_ntaps[0] : p.x6_nu);
    // (round 6, two f16 pieces: half the matrix work per tile, so the fixed costs weigh more -- launches with at most 64 tiles of the
    // large shape (the C = 768 units, 768 <-> 1536: 2048 or 1024 columns) take the 32-column wave tile whatever the reduction
    // length: K is cut into 4 slices instead of 8, half the partial sums to dump and re-read.  Measured per layer, RH_X6_TN=1
    // against the default: C = 768 k = 3 48.8 -> 41.6 us forward / 50.0 -> 42.8 data gradient, 768 -> 1536 57.3 -> 52.8 / 55.6 -> 53.2,
    // 1536 -> 768 61.1 -> 53.4 / 57.5 -> 54.2; the 128-tile launches (C = 384) lose with it and keep the 64-column tile)
    if (pl->tm >= 2 && (tn_env == 1 || (tn_env == 0 && ((blocks < 384 && total_steps <= 64) || blocks <= 64)))) {
        const int b1 = shape(1, wm0);
        if (b1 == 0) blocks = shape(2, wm0);
        else blocks = b1;
    }
    if (blocks == 0) return false;
    const int total_chunks = (p.C * is) >> 4;
    // K is split across workgroups when the output tiles alone cannot fill 256 CUs x 2.  Measured (layer table,
    // profiles/): 256-tile launches run faster UNSPLIT when the epilogue is a plain store (no partial sums to write and
    // re-read, no finalize launch), but slower when the epilogue reads the saved input for the activation derivative
    // (one round of workgroups exposes those loads; the finalize pass streams them) -- hence two thresholds.
    static const int split_env = [] { const char* e = getenv("RH_X6_SPLIT_BELOW"); return e ? atoi(e) : 0; }();
    static const int split_target = [] { const char* e = getenv("RH_X6_SPLIT_TARGET"); return e ? atoi(e) : 512; }();
    // (round 5: one threshold.  Rounds 2-4 split launches whose epilogue reads the saved input already below 384 workgroups --
    // measured per layer, one launch at a time; inside the step, beside the weight-gradient stream, the 256-tile data gradients (the k = 1
    // convs of the C = 384 units) run faster UNSPLIT: 10.01-10.03 ms per step against 10.05-10.09 in three alternated pairs,
    // profiles/round5_ab_knobs_and_negative_results.txt)
    const int split_below = split_env > 0 ? split_env : 200;
    int z = 1;
    if (blocks < split_below) {
        z = rh_cdiv(split_target, blocks);
        if (z > total_chunks / 2) z = total_chunks / 2;
        if (z > 16) z = 16;
        if (z < 1) z = 1;
    }
    pl->chunks_per_split = rh_cdiv(total_chunks, z);
    pl->ksplit = rh_cdiv(total_chunks, pl->chunks_per_split);
    p.part_stride = (long)p.B * p.Mr * p.out_row;
    // scratch of a split launch: whole tiles as they lie in the registers, [slice][tile] (x6_combine)
    // (or, for the finalize launch, in the output's layout: the larger of the two)
    pl->part_bytes = 0;
    if (pl->ksplit > 1) {
        const int64_t tiled = (int64_t)pl->ksplit * blocks * (32 * pl->tm * pl->wm) * (32 * pl->tn * pl->wn) * (int64_t)sizeof(float);
        const int64_t flat = (int64_t)pl->ksplit * p.part_stride * (int64_t)sizeof(float);
        pl->part_bytes = tiled > flat ? tiled : flat;
    }
    const unsigned long long in_b = 4ull * p.B * p.C * (unsigned long long)p.in_row;
    const unsigned long long row_span = (unsigned long long)p.Mr * (unsigned long long)p.out_row;
    const unsigned long long out_b = 4ull * (unsigned long long)p.part_stride;     // the epilogue's buffer descriptors
    return in_b < 0x7fffffffull && (unsigned long long)p.wq_bytes < 0x7fffffffull && row_span < 0x7fffffffull && out_b < 0x7fffffffull;
}

}  // namespace

// Tile fields of p / LDS bytes / grid for a FORCED tile shape without K split (unit_x6.hip: the fused residual unit needs
// one row tile per workgroup).  false = the shape does not fit (LDS, 2 GiB descriptors, task slots).
bool rh_conv_x6_plan_fixed(ConvP& p, int tm, int tn, int wm, size_t* lds, dim3* grid) {
    if (!x6_enabled() || p.x6_mode != 1 || p.is != 1 || p.inner != 1 || p.nphase != 1) return false;
    if (((uintptr_t)p.in & 3)) return false;
    if (RH_X6_F16 && !p.in_range) return false;
    p.vs = 0;
    p.Mr = p.M;
    const int wn = 4 / wm;
    const int BM = 32 * tm * wm, BN = 32 * tn * wn;
    if (BM != p.Mp) return false;
    int bnl = BN;
    if (p.ncols < BN) {
        bnl = 32;
        while (bnl < p.ncols) bnl <<= 1;
    }
    p.bnl = bnl;
    p.bnl_shift = __builtin_ctz(bnl);
    p.nb = BN / bnl;
    p.tiles_per_b = rh_cdiv(p.ncols, bnl);
    const int span = (p.ph_maxoff[0] - p.ph_minoff[0]) * p.inner;
    p.pitch = bnl + span;
    p.x6_P = p.nb * p.pitch;
    const int nq = wn * tn == 8 ? 3 : 2;
    if (2 * p.x6_P > 256 * nq) return false;
    p.x6_a_units = 2 * kX6P * BM;
    p.x6_b_units = 2 * kX6P * p.x6_P;
    *lds = (size_t)(2 * p.x6_a_units + 2 * p.x6_b_units) * 16;
    if (*lds > 160 * 1024) return false;
    p.ksplit = 1;
    p.chunks_per_split = p.C >> 4;
    p.part_stride = (long)p.B * p.Mr * p.out_row;
    const unsigned long long in_b = 4ull * p.B * p.C * (unsigned long long)p.in_row;
    const unsigned long long out_b = 4ull * (unsigned long long)p.part_stride;
    if (!(in_b < 0x7fffffffull && (unsigned long long)p.wq_bytes < 0x7fffffffull && out_b < 0x7fffffffull)) return false;
    *grid = dim3(rh_cdiv(p.B, p.nb) * p.tiles_per_b, 1, 1);
    return true;
}

int64_t rh_conv_x6_workspace(ConvP p) {
    X6Plan pl{};
    static const unsigned any_range[kRangeSlotWords] = {};
    if (!p.in_range) p.in_range = any_range;           // planning only: the answer does not depend on the slot
    if (!plan_x6(p, &pl)) return -1;
    if (pl.part_bytes > 0) (void)ticket_pool();        // (allocated outside any stream capture: the planning call comes first)
    return pl.part_bytes;
}

// Diagnostics (rh_conv1d_plan_info): the tile shape / split a launch of this geometry would get.
// out = {tm, tn, wm, ksplit, input stride of the fragment layout, vs, workgroups}; false = the geometry does not take this path.
bool rh_conv_x6_plan_query(ConvP p, int* out) {
    X6Plan pl{};
    static const unsigned any_range[kRangeSlotWords] = {};
    if (!p.in_range) p.in_range = any_range;
    if (!plan_x6(p, &pl)) return false;
    out[0] = pl.tm; out[1] = pl.tn; out[2] = pl.wm; out[3] = pl.ksplit;
    out[4] = p.is;
    out[5] = p.vs;
    out[6] = pl.col_tiles * pl.row_tiles * p.nphase * pl.ksplit;
    return true;
}

// Returns RH_OK with *used = false when the geometry (or the scratch offered) does not fit this path.
int rh_conv_launch_x6(ConvP& p, hipStream_t stream, const char* what, void* ws, int64_t ws_bytes, bool* used) {
    *used = false;
    ConvP q = p;
    X6Plan pl{};
    if (!plan_x6(q, &pl)) return RH_OK;
    if (pl.part_bytes > 0 && (!ws || ws_bytes < pl.part_bytes || ((uintptr_t)ws & 15) || pl.part_bytes >= 0x7fffffffll)) {     // no (usable) scratch offered: run unsplit
        pl.ksplit = 1;
        pl.chunks_per_split = (q.C * q.is) >> 4;
    }
    q.in_bytes = (unsigned)(4ull * q.B * q.C * (unsigned long long)q.in_row);
    q.w_range = q.wq + q.wq_bytes / 4;                 // the range record behind the fragments (conv_host.hip: fill_pack)
    q.tickets = nullptr;
    if (pl.ksplit > 1 && pl.ksplit <= combine_max()) {
        q.tickets = ticket_segment((unsigned)(pl.col_tiles * pl.row_tiles * q.nphase));
        RH_REQUIRE(q.tickets, RH_ERR_INVALID, "%s: no arrival counters for a K-split launch (hipMalloc failed, or first use under stream capture)", what);
    }
    q.part = (float*)ws;
    q.ksplit = pl.ksplit;
    q.chunks_per_split = pl.chunks_per_split;
    dim3 grid(pl.col_tiles, pl.row_tiles, q.nphase * q.ksplit);
    // the epilogue is a template parameter of the kernel; a K-split launch combines its slices itself (x6_combine) and the last
    // arriver of a tile runs the same epilogue
    const int epi = (q.ksplit > 1 && !q.tickets ? 0 : epi_mode(q)) | (q.vs == 2 ? 16 : (q.vs == 4 ? 32 : 0));
    const X6Launch go = x6_launcher(q.is, pl.tm, pl.tn, pl.wm);
    RH_REQUIRE(go && go(q, epi, grid, pl.lds, stream), RH_ERR_UNSUPPORTED, "%s: no conv_x6_kernel instance for stride %d tile %d%d%d epilogue %d",
               what, q.is, pl.tm, pl.tn, pl.wm, epi);
    if (int e = rh_check_launch(what)) return e;
    *used = true;
    if (q.ksplit > 1 && !q.tickets) {   // the partial sums are laid out like the output: finalize with the caller's (real-row) view
        ConvP f = p;
        f.part = q.part; f.ksplit = q.ksplit; f.part_stride = q.part_stride;
        f.out_range = q.out_range;                     // the finalize pass sees the final values: it publishes their max
        return rh_splitk_finalize_launch(f, stream);
    }
    return RH_OK;
}

// ---- stride-3 1-D gathers on the 2-D kernel (conv2d_x6.hip): the sequence is a plane of width 1, the stride a multiplier on
// the lane's patch position (3 is coprime with the 16 slots of a ds_read_b128 lane group: conflict-free).  Forward only in
// practice: bias / output LeakyReLU epilogues; anything with an input activation, a derivative or a residual operand stays
// on the f32 kernels.
namespace {
bool fill_c2x_from_1d(const ConvP& p, C2X* q) {
    if (p.x6_mode != 1 || p.is == 1 || p.is == 2 || p.is == 4 || p.inner != 1 || p.nphase != 1 || p.os != 1) return false;
    if (p.in_act != RH_ACT_NONE || p.epi_act != RH_ACT_NONE || p.mul_src || p.add || p.in_alpha || p.mul_alpha) return false;
    if (p.in_row != p.in_valid || p.out_row != p.out_valid || p.ph_ntaps[0] < 1 || p.ph_ntaps[0] > kMaxTaps) return false;
    *q = C2X{};
    q->in = p.in; q->wq = p.wq; q->out = p.out; q->bias = p.bias;
    q->B = p.B; q->C = p.C; q->M = p.M; q->Mp = p.Mp;
    q->in_h = p.in_row; q->in_w = 1; q->out_h = p.out_row; q->out_w = 1;
    q->rows = p.ncols; q->qcols = 1;
    q->is_h = p.is; q->is_w = 1; q->os_h = 1; q->os_w = 1;
    q->out_act = p.out_act; q->out_slope = p.out_slope;
    q->nphase = 1;
    q->ph_oph_h[0] = p.ph_oph[0]; q->ph_oph_w[0] = 0;
    q->ph_ntaps[0] = p.ph_ntaps[0]; q->ph_tap0[0] = 0;
    q->ph_minh[0] = p.ph_minoff[0]; q->ph_maxh[0] = p.ph_maxoff[0]; q->ph_minw[0] = q->ph_maxw[0] = 0;
    q->ph_q2ofs[0] = p.ph_q2ofs[0];
    for (int t = 0; t < p.ph_ntaps[0]; ++t) { q->offh[t] = p.off[p.ph_tap0[0] + t]; q->offw[t] = 0; }
    q->wq_bytes = p.wq_bytes;
    q->in_range = p.in_range; q->out_range = p.out_range;      // (f16 build; the weights' record sits behind the fragments of both packers)
    return true;
}
}  // namespace

bool rh_conv_c2x_query(ConvP p) {
    C2X q;
    long v[8];
    return x6_enabled() && fill_c2x_from_1d(p, &q) && rh_conv2d_x6_plan_query(q, v);
}

int rh_conv_launch_c2x(ConvP& p, hipStream_t stream, const char* what, bool* used) {
    *used = false;
    C2X q;
    if (!x6_enabled() || !fill_c2x_from_1d(p, &q)) return RH_OK;
    return rh_conv2d_x6_launch(q, stream, what, used);
}
