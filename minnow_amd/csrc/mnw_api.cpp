// mnw_api.cpp -- C ABI of libminnow_hip.so (include/minnow_hip.h): contexts, staging, launch
// scheduling and the exact path for order-dependent blocks.  Host side of the product; compiled by
// hipcc together with mnw_kernels.hip.  No CPU fallback: every compute entry point needs a HIP device.
#include "../../include/minnow_hip.h"
#include "mnw_common.h"

#include <math.h>
#include <stdio.h>
#include <string.h>
#include <stdlib.h>
#include <errno.h>
#include <ctype.h>
#include <locale.h>

#include <algorithm>
#include <condition_variable>
#include <mutex>
#include <string>
#include <thread>
#include <cerrno>
#include <unistd.h>
#include <sched.h>
#include <functional>
#include <vector>

namespace mnw {
void launch_prepare(hipStream_t, const void*, void*, int64_t, void*, int64_t, void*, int64_t, const void* = nullptr, void* = nullptr, int64_t = 0);
void launch_reduce_f32(hipStream_t, int, const DevJob*, int, int64_t, uint64_t*, void*, Meta*);
size_t job_sync_bytes();
void launch_reduce_i64(hipStream_t, int, const DevJob*, int, int64_t, uint64_t*);
void launch_pack_f32(hipStream_t, int, const DevJob*, int, int64_t, void*);
void launch_pack_i64(hipStream_t, int, const DevJob*, int, int64_t, const uint64_t*, Meta*);
void launch_encode_f32(hipStream_t, int, const DevJob*, int, const int64_t*, const int*, int, uint64_t*, void*, uint32_t*, Meta*);
int encode_f32_blocks_per_cu();
void launch_encode_reg_f32(hipStream_t, int, const DevJob*, int, int64_t, const int*, void*, void*, uint32_t, uint32_t*, uint32_t*, Meta*, int, int64_t, int, int, void*, int, int);
int encode_reg_f32_sweepers();
void launch_encode_reg_i64(hipStream_t, int, const DevJob*, int, int64_t, const int*, void*, uint32_t, uint32_t*, uint32_t*, Meta*, int, int64_t, void*, int, int);
int64_t reg_item_elems_i64();
int encode_reg_i64_blocks_per_cu();
int encode_reg_i64_sweepers();
size_t item_rec_bytes();
int reg_max_items_per_block();
int reg_max_items_per_big_block();
int reg_group_items();
int encode_reg_f32_xf_blocks_per_cu();
int64_t reg_item_elems();
int encode_reg_f32_blocks_per_cu();
void launch_unpack_f32(hipStream_t, int, bool, bool, const DevJob*, int, int64_t);
void launch_unpack_i64(hipStream_t, int, const DevJob*, int, int64_t);
void launch_subcell_gather(hipStream_t, int, const float*, float*, int64_t, int64_t);
void launch_subcell_scatter(hipStream_t, int, const float*, float*, int64_t, int64_t, int, float);
void launch_bounds3(hipStream_t, int, const float*, int64_t, uint32_t*);
void launch_reduce_vec3(hipStream_t, int, const float*, int64_t, int64_t, const DevJob*, uint64_t*);
void launch_pack_vec3(hipStream_t, int, const float*, int64_t, int64_t, const DevJob*, const uint64_t*, Meta*);
int64_t vec3_tiles(int64_t, int64_t);
int64_t vec3_reg_recs(int64_t, int64_t);
int64_t vec3_reg_max_tiles_per_block();
int vec3_reg_sweepers();
void vec3_reg_set_sweepers(int);
bool vec3_reg_geometry_ok(int64_t, int64_t);
int encode_reg_vec3_blocks_per_cu(int);
void launch_encode_reg_vec3(hipStream_t, int, const float*, int64_t, int64_t, const DevJob*, void*, uint32_t, uint32_t*, uint32_t*, Meta*, int64_t, int, int);
void launch_unpack_vec3(hipStream_t, int, float*, int64_t, int64_t, const DevJob*, int, float, int, bool, int, int);
bool vec3_unpack_geometry_ok(int64_t, int64_t);
constexpr int kMinhBatch = 64;
struct MinhCol { const float* src; float* dst; int64_t n; float low, high, high_below; int32_t xform; };
struct MinhArgs { MinhCol col[kMinhBatch]; };
void launch_minh_process(hipStream_t, int, const MinhArgs&, int, int64_t);
void launch_minh_unlog(hipStream_t, int, const MinhArgs&, int, int64_t);
void launch_minh_selftest(hipStream_t, int, uint64_t, uint64_t, unsigned long long*);
constexpr int kTextTile = 12288, kTextMaxCols = 256, kTextMaxOut = 128;
struct TextCols { int16_t slot[kTextMaxCols]; void* out[kTextMaxOut]; int32_t n_float, n_int; int32_t fields_expected; int32_t max_field; };
struct TextSlow { int64_t row; int64_t offset; int32_t len; int32_t slot; };
struct TextState { unsigned long long err_line; unsigned int n_slow; unsigned int flags; long long n_rows; long long n_lines; unsigned int tiles_general; unsigned int pad; };
void launch_text_count(hipStream_t, const uint8_t*, int64_t, uint8_t, uint8_t, int64_t*);
void launch_text_scan(hipStream_t, int64_t*, int64_t, TextState*);
void launch_text_parse(hipStream_t, const uint8_t*, int64_t, uint8_t, uint8_t, const int64_t*, const TextCols&, int64_t, TextSlow*, unsigned, TextState*);
void launch_text_patch(hipStream_t, const TextSlow*, unsigned, const TextCols&);
constexpr int kCutTile = 2048, kCutMaxCols = 64;
struct CutCols { const void* src[kCutMaxCols]; void* dst[kCutMaxCols]; uint8_t esz[kCutMaxCols]; int32_t n; };
void launch_cut_count(hipStream_t, const float*, float, int64_t, int64_t*);
void launch_cut_scatter(hipStream_t, const float*, float, int64_t, const int64_t*, const CutCols&);
void launch_quantize_i64(hipStream_t, int, const float*, int64_t, float, float, int64_t*);
void launch_periodic_min_seq(hipStream_t, const int64_t*, int64_t, int64_t, int64_t, int64_t*);
void launch_bound_i64(hipStream_t, int, int64_t*, int64_t, int64_t, int64_t);
}  // namespace mnw

using mnw::DevJob;
using mnw::Meta;

static_assert(sizeof(Meta) == sizeof(mnw_block_meta), "meta layout");

struct TimedLaunch {
    int kernel;
    hipEvent_t start, stop;
};

constexpr int kSlots = 3;
constexpr int kSyncPad = 32;      // keeps the abort flag on its own 128-byte line
struct JobSlot {
    int cap = 0;
    // upload block: one pinned host allocation and its device mirror with the same layout
    // [DevJob x cap][gtile: int64 x (cap+1)][gjob: int x (cap+2)]; k_prepare copies it in one launch
    uint8_t* h_up = nullptr;
    uint8_t* d_up = nullptr;
    size_t up_bytes = 0;
    DevJob* d_jobs = nullptr;
    DevJob* h_jobs = nullptr;
    int64_t* d_gtile = nullptr;   // fused encode: group boundaries in the batch-wide tile numbering
    int64_t* h_gtile = nullptr;
    int* d_gjob = nullptr;        // fused encode: first job of each group
    int* h_gjob = nullptr;
    // work block (device): [stats x cap][abort pad][JobSync x cap]; k_prepare zeroes the used parts
    uint64_t* d_stats = nullptr;
    uint8_t* d_sync = nullptr;    // kSyncPad words (the first is the fused launch's abort flag), then one
                                  // mnw::JobSync (arrival count + published descriptor word) per job
    size_t sync_stride = 64;
    // results: pinned host memory the kernels write directly (no device-to-host copy)
    Meta* h_meta = nullptr;
    Meta* d_meta = nullptr;       // device address of h_meta
    uint32_t* h_sync = nullptr;   // abort flag of the fused launch, copied here by a 1-word k_prepare
    uint32_t* d_hsync = nullptr;  // device address of h_sync
    // single-read encode: the job of every item (pinned, coherent) and its device copy, grow-only
    int* h_itemjob = nullptr;
    int* d_itemjob = nullptr;
    size_t cap_itemjob = 0;
    // What the device copy of the tables holds, if it was uploaded by a path that can reuse it (decode,
    // single-read encode): a batch with byte-identical tables -- a pipeline that cycles through the same
    // buffers -- skips the k_prepare launch (~7 us: it reads pinned host memory across the link).
    std::vector<uint8_t> shadow;
    int shadow_tag = 0;           // 0: nothing reusable, 1: decode tables, 2: single-read encode tables
    hipEvent_t consumed = nullptr;
    bool in_flight = false;
};

// ---- host <-> device staging for the host-pointer entry points (what the cgo shim binds: Go slices, pageable memory) -----
// A pageable hipMemcpy runs at 7-9 GB/s on this platform (the runtime stages it through small internal buffers, one thread).
// Here the bytes go through library-owned PINNED buffers in chunks: a few host threads copy chunk k+1 from the caller's memory
// into a pinned buffer while the DMA engine moves chunk k, and the way back mirrors it (SURVEY.md 8b "Ownership": sync calls
// retain nothing of the caller's memory after they return; library-owned pinned host buffers).
struct CopyPool {                  // a fixed set of threads that split one memcpy between them
    std::vector<std::thread> th;
    std::mutex mu;
    std::condition_variable cv_go, cv_done;
    uint8_t* dst = nullptr; const uint8_t* src = nullptr; size_t n = 0;
    int parts = 0, next_part = 0, done_parts = 0;
    uint64_t epoch = 0;
    bool quit = false;
    const std::function<void(int)>* task = nullptr;     // set: part k runs (*task)(k); else it is a slice of one memcpy
    void part(int k) {
        if (task) { (*task)(k); return; }
        const size_t per = ((n + parts - 1) / parts + 4095) & ~(size_t)4095;
        const size_t a = per * k, b = a + per < n ? a + per : n;
        if (a < b) memcpy(dst + a, src + a, b - a);
    }
    void worker() {
        uint64_t seen = 0;
        std::unique_lock<std::mutex> lk(mu);
        for (;;) {
            cv_go.wait(lk, [&] { return quit || epoch != seen; });
            if (quit) return;
            seen = epoch;
            while (next_part < parts) {
                const int k = next_part++;
                lk.unlock(); part(k); lk.lock();
                if (++done_parts == parts) cv_done.notify_all();
            }
        }
    }
    void start(int threads) { for (int i = 0; i < threads; ++i) th.emplace_back([this] { worker(); }); }
    // nparts independent pieces of work, fn(k) for k in [0, nparts), shared between the pool's threads and the caller
    void run(int nparts, const std::function<void(int)>& fn) {
        if (nparts <= 0) return;
        if (th.empty() || nparts == 1) { for (int k = 0; k < nparts; ++k) fn(k); return; }
        std::unique_lock<std::mutex> lk(mu);
        task = &fn;
        parts = nparts; next_part = 0; done_parts = 0; ++epoch;
        cv_go.notify_all();
        while (next_part < parts) {
            const int k = next_part++;
            lk.unlock(); part(k); lk.lock();
            ++done_parts;
        }
        cv_done.wait(lk, [&] { return done_parts == parts; });
        task = nullptr;
    }
    void copy(void* d, const void* s2, size_t bytes) {
        if (bytes < (1u << 20) || th.empty()) { memcpy(d, s2, bytes); return; }
        std::unique_lock<std::mutex> lk(mu);
        task = nullptr;
        dst = (uint8_t*)d; src = (const uint8_t*)s2; n = bytes;
        parts = (int)th.size() + 1; next_part = 0; done_parts = 0; ++epoch;
        cv_go.notify_all();
        while (next_part < parts) {          // the calling thread takes parts too
            const int k = next_part++;
            lk.unlock(); part(k); lk.lock();
            ++done_parts;
        }
        cv_done.wait(lk, [&] { return done_parts == parts; });
    }
    ~CopyPool() {
        { std::lock_guard<std::mutex> g(mu); quit = true; }
        cv_go.notify_all();
        for (auto& t : th) t.join();
    }
};

struct Stager {
    static constexpr int kBufs = 4;
    size_t chunk = 16u << 20;
    uint8_t* pin[kBufs] = {nullptr, nullptr, nullptr, nullptr};
    hipEvent_t ev[kBufs] = {nullptr, nullptr, nullptr, nullptr};
    bool busy[kBufs] = {false, false, false, false};
    int next = 0;
    CopyPool* pool = nullptr;
    bool ready = false;
};

struct mnw_ctx {
    int device = 0;
    int cus = 256;
    hipStream_t stream = nullptr;
    std::string err;

    // job / stats / meta arrays (grow-only, like bit.ArrayBuffer, bit.go:183-205).  A ring of
    // slots so that a batch can be prepared while the kernels of earlier batches still read theirs;
    // a slot is reused only after the event recorded behind its last consumer has completed.
    JobSlot slots[kSlots];
    JobSlot dslots[kSlots];       // decode batches cycle through their own ring (their tables repeat among themselves)
    int cur = 0, dcur = 0;
    JobSlot* pend = nullptr;      // slot of the pending async encode batch

    // staging for the host-pointer entry points
    Stager stg;
    int host_threads = 0;          // threads that copy between the caller's memory and the pinned buffers (0: min(8, cores))
    int64_t stage_min = 4 << 20;   // transfers below this many bytes go through a plain hipMemcpy
    size_t cap_in = 0, cap_out = 0;
    void* d_in = nullptr;
    void* d_out = nullptr;

    // minp vector blocks: structure-of-arrays staging [3][n_file^3] between the layout pass and the codec
    size_t cap_soa = 0;
    float* d_soa = nullptr;
    uint32_t* d_bkeys = nullptr;  // 8 words: bounds keys
    uint32_t* h_bkeys = nullptr;  // pinned

    // minh value transforms for the encode paths that do not apply them at load time (two-pass, fused, exact):
    // the processed copy of the block, like minh.buf (go/minh/minh.go:128-133)
    size_t cap_xf = 0;
    float* d_xf = nullptr;

    // text ingest scratch (grow-only): the text, line starts, kept-line flags, per-tile counts and bases, slow list
    size_t cap_text = 0, cap_tile = 0;
    void* d_text = nullptr; void* d_tile = nullptr;
    const void* d_text_now = nullptr;               // where the resident block's bytes are: d_text, or a device buffer the caller uploaded (mnw_text_rows_resident)
    void* d_tslow = nullptr; void* d_tstate = nullptr;
    size_t cap_tslow = 0;                           // literals the slow list holds (grows when a block needs more)
    const uint8_t* text_res = nullptr;              // host text whose copy, tile counts and prefixes are on the device (mnw_text_rows)
    int64_t text_res_n = 0, text_res_lines = 0, text_res_rows = 0;
    uint8_t text_res_sep = 0, text_res_comm = 0;
    int64_t text_stats[3] = {0, 0, 0};              // of the last mnw_text_parse: lines, lines kept, tiles on the general path

    // exact path scratch
    size_t cap_q = 0;
    int64_t* d_q = nullptr;
    int64_t* d_state = nullptr;

    // pending async encode batch
    bool pending = false;
    std::vector<mnw_enc_job> pend_jobs;
    std::vector<int> pend_slot;    // caller index -> device job slot (-1: handled on the host)
    std::vector<Meta> pend_host;   // host-resolved metas
    hipStream_t pend_stream = nullptr;

    // options
    int64_t group_bytes = 1ll << 40;   // two-kernel path: all blocks of a batch in one reduce + one pack launch
    int grid_cap = 0;
    int fused = 0;               // float encode as one persistent launch per batch (k_encode_f32): opt-in, see DESIGN.md
    int64_t reg_drop_item = -1;  // TEST HOOK: the single-read kernel never publishes this item's record -> bounded wait, abort, host fallback
    int reg_weighted = 1;        // single read: weighted item hand-out (see k_encode_reg_f32)
    int vec3_fused = 1;          // minp vectors: reduce + pack straight from the AoS grid when the geometry allows
    int single_read = 1;         // float encode through k_encode_reg_f32 (block read once) whenever the batch is eligible
    int vec3_tiles_per_wg = 4;   // minp vector kernels: vtiles per workgroup (0: persistent grids of 8 / 4 workgroups per CU).  Like the
                                 // decode launches they run faster as many short workgroups: reduce 315 -> 300 us, pack 482 -> 437, unpack 673 -> 646
                                 // on a 500^3 file cell (tools/ab_minp.py)
    int big_blocks = 1;          // ... including blocks of more than 512 items (two-level record sweep), up to twice the grid
    int reg_blocks_per_cu = 0;
    int reg_xf_blocks_per_cu = 0;  // ... of its instantiation with the minh value transform   // co-resident workgroups of k_encode_reg_f32 per CU (occupancy query)
    int reg_vec3_blocks_per_cu = 0; // co-resident workgroups of k_encode_reg_vec3 per CU
    int single_read_backoff = 1; // after a give-up, the next batches skip the single-read kernels (1, 2, 4, ... 64 batches on consecutive give-ups)
    int sr_backoff = 0, sr_backoff_left = 0;
    int reg_round_table = 0;     // which of the kernels' round tables the weighted hand-out uses (0 = 14/12/11/10; the others: measurements)
    int unlog_fused = 1;         // minh reader: 10^x applied inside k_unpack_f32 (one write of the column) rather than by k_minh_unlog afterwards
    int float_flat_items = 512;  // float blocks of more items (of 8,192) get their descriptor from the sweeper workgroups of k_encode_reg_f32<., BIG>
    int int_flat_items = 256;    // int64 blocks of more items (of 4,096) get their descriptor from the sweeper workgroups of k_encode_reg_i64 (2^21-element blocks: 0.60 of 8 TB/s against 0.52 through the flat sweep of 512 records per workgroup and trip; at 256 items the two are equal)
    int vec3_unpack_tiles = 8;   // minp decode: vtiles per workgroup of k_unpack_vec3 with vec3_x_order (505 us at 8 against 521-536 at 4 and 529 at 16)
    int vec3_x_order = 2;        // minp decode: 1 = workgroups in dispatch order take the sub-cells of one x-row together; 2 = ... the workgroups of one XCD do (one L2 sees a whole row of the grid)
    int vec3_unpack_db = 1;      // k_unpack_vec3: two LDS word buffers and one barrier per vtile (0: one buffer, two barriers, more workgroups per CU)
    int vec3_deep = 4;           // k_encode_reg_vec3: items per workgroup between a vtile's publication and its pack (3: four workgroups per CU; 4, 5: three)
    int vec3_round_table = 4;    // k_encode_reg_vec3 (depth 4, 5): 0 = every round a whole grid, k > 0 = weighted hand-out by thirds of the grid, table k - 1 (4: 10 / 9 / 8 items per cycle)
    int vec3_single_read = 1;    // minp vectors: the grid is read once (k_encode_reg_vec3) when the geometry allows and its grid can be co-resident;
                                 // 0: k_reduce_vec3 + k_pack_vec3, which read it twice (also what reruns the call when the single-read launch gives up)
    int reg_i64_blocks_per_cu = 0; // co-resident workgroups of k_encode_reg_i64 per CU
    int single_read_int = 1;     // int64 encode through k_encode_reg_i64 whenever the batch's int blocks are eligible
    bool pend_int_reg = false;   // the pending batch's int jobs went through k_encode_reg_i64
    int pend_nslot = 0;          // device slots of the pending batch (float jobs, then int jobs)
    int pend_mode = 0;           // 0 two-kernel, 1 fused, 2 single read
    void* d_recs = nullptr;      // single read: one published statistics record per item (grow-only)
    size_t cap_recs = 0;
    uint32_t reg_gen = 0;        // generation tag of the records of the current batch (never 0)
    int fused_blocks_per_cu = 0; // co-resident workgroups of k_encode_f32 per CU (occupancy query)
    int64_t counters[MNW_C_COUNT] = {0, 0, 0, 0, 0, 0, 0, 0};   // mnw_counters_get
    int pend_nfloat = 0;         // device slots [0, pend_nfloat) of the pending batch are float jobs
    bool pend_fused = false;

    // timing
    bool timing = false;
    std::vector<TimedLaunch> launches;
    std::vector<hipEvent_t> event_pool;
    mnw_kernel_time acc[MNW_K_COUNT];
};

namespace {

int fail(mnw_ctx* c, int status, const char* what, hipError_t e = hipSuccess) {
    if (c) {
        char buf[512];
        if (e != hipSuccess) snprintf(buf, sizeof buf, "%s: %s", what, hipGetErrorString(e));
        else snprintf(buf, sizeof buf, "%s", what);
        c->err = buf;
    }
    return status;
}

#define HIPCHK(c, call)                                                          \
    do {                                                                         \
        hipError_t e_ = (call);                                                  \
        if (e_ != hipSuccess) return fail((c), MNW_E_HIP, #call, e_);            \
    } while (0)

void free_slot(JobSlot& sl) {
    if (sl.d_up) {
        hipFree(sl.d_up); hipHostFree(sl.h_up); hipFree(sl.d_stats); hipHostFree(sl.h_meta); hipHostFree(sl.h_sync);
    }
    sl.d_up = nullptr; sl.d_jobs = nullptr; sl.cap = 0;
    if (sl.d_itemjob) { hipFree(sl.d_itemjob); hipHostFree(sl.h_itemjob); }
    sl.d_itemjob = nullptr; sl.h_itemjob = nullptr; sl.cap_itemjob = 0;
    sl.shadow_tag = 0;
}

// next ring slot with room for n jobs; waits (host side) only if that slot's previous batch is
// still being consumed by the GPU
int acquire_slot(mnw_ctx* c, int n, JobSlot** out, bool decode_ring = false) {
    int& cur = decode_ring ? c->dcur : c->cur;
    JobSlot* ring = decode_ring ? c->dslots : c->slots;
    cur = (cur + 1) % kSlots;
    JobSlot& sl = ring[cur];
    if (&sl == c->pend) { cur = (cur + 1) % kSlots; return acquire_slot(c, n, out, decode_ring); }
    if (!sl.consumed) HIPCHK(c, hipEventCreateWithFlags(&sl.consumed, hipEventDisableTiming));
    if (sl.in_flight) { HIPCHK(c, hipEventSynchronize(sl.consumed)); sl.in_flight = false; }
    if (n > sl.cap) {
        int cap = sl.cap ? sl.cap : 64;
        while (cap < n) cap *= 2;
        free_slot(sl);
        const size_t off_gtile = sizeof(DevJob) * cap;
        const size_t off_gjob = off_gtile + sizeof(int64_t) * (cap + 1);
        sl.up_bytes = (off_gjob + sizeof(int) * (cap + 2) + 15) / 16 * 16;
        // Coherent (fine-grained) pinned memory: the GPU reads the job table straight out of it in
        // k_prepare and writes the metadata straight into it, batch after batch.  With the default
        // (non-coherent) kind the GPU may keep lines of it cached across launches and copy a stale table.
        HIPCHK(c, hipHostMalloc((void**)&sl.h_up, sl.up_bytes, hipHostMallocCoherent | hipHostMallocMapped));
        HIPCHK(c, hipMalloc((void**)&sl.d_up, sl.up_bytes));
        memset(sl.h_up, 0, sl.up_bytes);
        sl.h_jobs = (DevJob*)sl.h_up; sl.d_jobs = (DevJob*)sl.d_up;
        sl.h_gtile = (int64_t*)(sl.h_up + off_gtile); sl.d_gtile = (int64_t*)(sl.d_up + off_gtile);
        sl.h_gjob = (int*)(sl.h_up + off_gjob); sl.d_gjob = (int*)(sl.d_up + off_gjob);
        sl.sync_stride = mnw::job_sync_bytes();
        const size_t stats_bytes = sizeof(uint64_t) * mnw::kStatSlots * mnw::kStatWords * cap;
        HIPCHK(c, hipMalloc((void**)&sl.d_stats, stats_bytes + sizeof(uint32_t) * kSyncPad + sl.sync_stride * cap));
        sl.d_sync = (uint8_t*)sl.d_stats + stats_bytes;
        HIPCHK(c, hipHostMalloc((void**)&sl.h_meta, sizeof(Meta) * cap, hipHostMallocCoherent | hipHostMallocMapped));
        HIPCHK(c, hipHostMalloc((void**)&sl.h_sync, sizeof(uint32_t) * kSyncPad, hipHostMallocCoherent | hipHostMallocMapped));
        HIPCHK(c, hipHostGetDevicePointer((void**)&sl.d_meta, sl.h_meta, 0));
        HIPCHK(c, hipHostGetDevicePointer((void**)&sl.d_hsync, sl.h_sync, 0));
        sl.cap = cap;
    }
    *out = &sl;
    return MNW_OK;
}

uint32_t* abort_ptr(JobSlot* sl) { return (uint32_t*)sl->d_sync; }
void* sync_ptr(JobSlot* sl, int j) { return sl->d_sync + sizeof(uint32_t) * kSyncPad + sl->sync_stride * j; }

// One small launch sets a batch up: job records (and group tables) from pinned host memory into
// device memory, statistics and synchronisation words of the first njobs jobs zeroed.
bool tables_unchanged(const JobSlot* sl, int tag, size_t bytes) {
    return sl->shadow_tag == tag && sl->shadow.size() == bytes && memcmp(sl->shadow.data(), sl->h_up, bytes) == 0;
}
void remember_tables(JobSlot* sl, int tag, size_t bytes) {
    sl->shadow.assign(sl->h_up, sl->h_up + bytes);
    sl->shadow_tag = tag;
}

int prepare_slot(mnw_ctx* c, JobSlot* sl, hipStream_t s, int njobs, bool zero) {
    sl->shadow_tag = 0;
    mnw::launch_prepare(s, sl->h_up, sl->d_up, (int64_t)sl->up_bytes,
                        zero ? sl->d_stats : nullptr, zero ? (int64_t)(sizeof(uint64_t) * mnw::kStatSlots * mnw::kStatWords * njobs) : 0,
                        zero ? sl->d_sync : nullptr, zero ? (int64_t)(sizeof(uint32_t) * kSyncPad + sl->sync_stride * njobs) : 0);
    HIPCHK(c, hipGetLastError());
    return MNW_OK;
}

int release_slot(mnw_ctx* c, JobSlot* sl, hipStream_t s) {
    HIPCHK(c, hipEventRecord(sl->consumed, s));
    sl->in_flight = true;
    return MNW_OK;
}

int ensure_dev(mnw_ctx* c, void** p, size_t* cap, size_t need) {
    if (need <= *cap) return MNW_OK;
    if (*p) hipFree(*p);
    *p = nullptr; *cap = 0;
    size_t want = need + need / 4 + 256;
    HIPCHK(c, hipMalloc(p, want));
    *cap = want;
    return MNW_OK;
}

int stager_init(mnw_ctx* c) {
    Stager& g = c->stg;
    if (g.ready) return MNW_OK;
    for (int b = 0; b < Stager::kBufs; ++b) {
        HIPCHK(c, hipHostMalloc((void**)&g.pin[b], g.chunk, hipHostMallocDefault));
        HIPCHK(c, hipEventCreateWithFlags(&g.ev[b], hipEventDisableTiming));
    }
    int t = c->host_threads;
    if (t <= 0) { const char* e = getenv("MNW_HOST_THREADS"); if (e) t = atoi(e); }      // (for contexts a caller does not create itself)
    if (t <= 0) {
        // the CPUs this process may run on, not the machine's (a container's share is what counts)
        unsigned hw = std::thread::hardware_concurrency();
        cpu_set_t set;
        if (sched_getaffinity(0, sizeof set, &set) == 0) { const int n = CPU_COUNT(&set); if (n > 0 && (unsigned)n < hw) hw = (unsigned)n; }
        t = hw >= 16 ? 8 : (hw >= 4 ? (int)hw / 2 : 1);
    }
    g.pool = new CopyPool();
    g.pool->start(t - 1);
    g.ready = true;
    return MNW_OK;
}

void stager_free(mnw_ctx* c) {
    Stager& g = c->stg;
    delete g.pool; g.pool = nullptr;
    for (int b = 0; b < Stager::kBufs; ++b) {
        if (g.pin[b]) hipHostFree(g.pin[b]);
        if (g.ev[b]) hipEventDestroy(g.ev[b]);
        g.pin[b] = nullptr; g.ev[b] = nullptr;
    }
    g.ready = false;
}

// Host (pageable) -> device, enqueued on s.  When it returns every byte has left the caller's buffer (it sits in device
// memory or in a pinned buffer whose DMA is queued on s): nothing of `src` is retained.
// the piece a staged transfer moves at a time: a whole pinned buffer, or -- so that a transfer of less than a few buffers still
// overlaps its host copy with its DMA -- a quarter of the transfer (1 MiB at least)
static size_t stage_step(const Stager& g, size_t bytes) {
    if (bytes >= 4 * g.chunk) return g.chunk;
    size_t st = ((bytes + 3) / 4 + 4095) & ~(size_t)4095;
    if (st < (1u << 20)) st = 1u << 20;
    return st < g.chunk ? st : g.chunk;
}

int copy_h2d(mnw_ctx* c, hipStream_t s, void* dev_dst, const void* src, size_t bytes) {
    if (bytes == 0) return MNW_OK;
    if ((int64_t)bytes < c->stage_min) {
        HIPCHK(c, hipMemcpyAsync(dev_dst, src, bytes, hipMemcpyHostToDevice, s));
        HIPCHK(c, hipStreamSynchronize(s));       // (a pageable source: the runtime may still be reading it)
        return MNW_OK;
    }
    int rc = stager_init(c);
    if (rc) return rc;
    Stager& g = c->stg;
    const size_t step = stage_step(g, bytes);
    for (size_t off = 0; off < bytes; off += step) {
        const size_t len = bytes - off < step ? bytes - off : step;
        const int b = g.next; g.next = (g.next + 1) % Stager::kBufs;
        if (g.busy[b]) { HIPCHK(c, hipEventSynchronize(g.ev[b])); g.busy[b] = false; }
        g.pool->copy(g.pin[b], (const uint8_t*)src + off, len);
        HIPCHK(c, hipMemcpyAsync((uint8_t*)dev_dst + off, g.pin[b], len, hipMemcpyHostToDevice, s));
        HIPCHK(c, hipEventRecord(g.ev[b], s));
        g.busy[b] = true;
    }
    return MNW_OK;
}

// n ranges of a file -> one contiguous device buffer (in the order given), through the pinned ring: the host threads pread the
// pieces of chunk k+1 straight into a pinned buffer while the DMA engine moves chunk k -- no pageable buffer in between, no
// page faults, no second host copy.
int copy_file_h2d_impl(mnw_ctx* c, hipStream_t s, void* dev_dst, int fd, int n, const int64_t* file_off, const int64_t* len, const int64_t* dst_off) {
    size_t total = 0;
    for (int i = 0; i < n; ++i) total += (size_t)len[i];
    if (total == 0) return MNW_OK;
    int rc = stager_init(c);
    if (rc) return rc;
    Stager& g = c->stg;
    struct Piece { int64_t foff; size_t len, at, doff; };
    const size_t kSub = 1u << 20;                             // pieces are cut to 1 MiB so that the threads share a chunk evenly
    std::vector<Piece> cur;
    size_t fill = 0;
    std::string err;
    std::mutex emu;
    auto flush = [&]() -> int {
        if (cur.empty()) return MNW_OK;
        const int b = g.next; g.next = (g.next + 1) % Stager::kBufs;
        if (g.busy[b]) { HIPCHK(c, hipEventSynchronize(g.ev[b])); g.busy[b] = false; }
        uint8_t* const pin = g.pin[b];
        const std::function<void(int)> fn = [&](int k) {
            const Piece& p = cur[(size_t)k];
            size_t got = 0;
            while (got < p.len) {
                const ssize_t r = pread(fd, pin + p.at + got, p.len - got, (off_t)(p.foff + (int64_t)got));
                if (r <= 0) { std::lock_guard<std::mutex> lk(emu); err = r == 0 ? "unexpected end of file" : strerror(errno); return; }
                got += (size_t)r;
            }
        };
        g.pool->run((int)cur.size(), fn);
        if (!err.empty()) return fail(c, MNW_E_IO, ("copy_file_to_device: " + err).c_str());
        // one DMA per run of pieces that are contiguous in the destination as well (they are in the pinned buffer by construction)
        for (size_t k = 0; k < cur.size();) {
            size_t k2 = k + 1, run = cur[k].len;
            while (k2 < cur.size() && cur[k2].doff == cur[k].doff + run) { run += cur[k2].len; ++k2; }
            HIPCHK(c, hipMemcpyAsync((uint8_t*)dev_dst + cur[k].doff, pin + cur[k].at, run, hipMemcpyHostToDevice, s));
            k = k2;
        }
        HIPCHK(c, hipEventRecord(g.ev[b], s));
        g.busy[b] = true;
        fill = 0; cur.clear();
        return MNW_OK;
    };
    const size_t step = stage_step(g, total);
    size_t packed = 0;
    for (int i = 0; i < n; ++i) {
        const size_t d0 = dst_off ? (size_t)dst_off[i] : packed;
        size_t off = 0;
        while (off < (size_t)len[i]) {
            size_t take = std::min((size_t)len[i] - off, step - fill);
            if (take > kSub) take = kSub;
            cur.push_back(Piece{file_off[i] + (int64_t)off, take, fill, d0 + off});
            off += take; fill += take;
            if (fill == step) { rc = flush(); if (rc) return rc; }
        }
        packed += (size_t)len[i];
    }
    return flush();
}

// n device -> host copies, pipelined through the pinned ring: every pinned chunk is filled by as many pieces as fit, the DMA of
// chunk k+1 .. k+3 runs while the host threads copy chunk k out to the callers' buffers.
int copy_d2h_batch_impl(mnw_ctx* c, hipStream_t s, int n, void* const* dst, const void* const* src, const int64_t* len) {
    size_t total = 0;
    for (int i = 0; i < n; ++i) total += (size_t)len[i];
    if (total == 0) return MNW_OK;
    int rc = stager_init(c);
    if (rc) return rc;
    Stager& g = c->stg;
    for (int b = 0; b < Stager::kBufs; ++b)
        if (g.busy[b]) { HIPCHK(c, hipEventSynchronize(g.ev[b])); g.busy[b] = false; }
    struct Piece { int item; size_t off, len, at; };
    std::vector<std::vector<Piece>> chunks;                  // the pieces of every pinned chunk
    {
        const size_t step = stage_step(g, total);
        std::vector<Piece> cur;
        size_t fill = 0;
        for (int i = 0; i < n; ++i) {
            size_t off = 0;
            while (off < (size_t)len[i]) {
                const size_t take = std::min((size_t)len[i] - off, step - fill);
                cur.push_back(Piece{i, off, take, fill});
                off += take; fill += take;
                if (fill == step) { chunks.push_back(cur); cur.clear(); fill = 0; }
            }
        }
        if (!cur.empty()) chunks.push_back(cur);
    }
    auto issue = [&](size_t k) -> hipError_t {
        const int b = (int)(k % Stager::kBufs);
        for (const Piece& p : chunks[k]) {
            const hipError_t e = hipMemcpyAsync(g.pin[b] + p.at, (const uint8_t*)src[p.item] + p.off, p.len, hipMemcpyDeviceToHost, s);
            if (e != hipSuccess) return e;
        }
        return hipEventRecord(g.ev[b], s);
    };
    for (size_t k = 0; k < chunks.size() && k < (size_t)Stager::kBufs; ++k) HIPCHK(c, issue(k));
    for (size_t k = 0; k < chunks.size(); ++k) {
        const int b = (int)(k % Stager::kBufs);
        HIPCHK(c, hipEventSynchronize(g.ev[b]));
        for (const Piece& p : chunks[k]) g.pool->copy((uint8_t*)dst[p.item] + p.off, g.pin[b] + p.at, p.len);
        if (k + Stager::kBufs < chunks.size()) HIPCHK(c, issue(k + Stager::kBufs));
    }
    g.next = 0;
    return MNW_OK;
}

// Device -> host (pageable), behind everything already enqueued on s; returns when the bytes are in `dst`.
int copy_d2h_impl(mnw_ctx* c, hipStream_t s, void* dst, const void* dev_src, size_t bytes) {
    if (bytes == 0) return MNW_OK;
    if ((int64_t)bytes < c->stage_min) {
        HIPCHK(c, hipMemcpyAsync(dst, dev_src, bytes, hipMemcpyDeviceToHost, s));
        HIPCHK(c, hipStreamSynchronize(s));
        return MNW_OK;
    }
    int rc = stager_init(c);
    if (rc) return rc;
    Stager& g = c->stg;
    for (int b = 0; b < Stager::kBufs; ++b)
        if (g.busy[b]) { HIPCHK(c, hipEventSynchronize(g.ev[b])); g.busy[b] = false; }
    const size_t step = stage_step(g, bytes);
    const size_t nchunks = (bytes + step - 1) / step;
    auto issue = [&](size_t k) -> hipError_t {
        const size_t off = k * step, len = bytes - off < step ? bytes - off : step;
        const int b = (int)(k % Stager::kBufs);
        hipError_t e = hipMemcpyAsync(g.pin[b], (const uint8_t*)dev_src + off, len, hipMemcpyDeviceToHost, s);
        if (e == hipSuccess) e = hipEventRecord(g.ev[b], s);
        return e;
    };
    for (size_t k = 0; k < nchunks && k < (size_t)Stager::kBufs; ++k) HIPCHK(c, issue(k));
    for (size_t k = 0; k < nchunks; ++k) {
        const size_t off = k * step, len = bytes - off < step ? bytes - off : step;
        const int b = (int)(k % Stager::kBufs);
        HIPCHK(c, hipEventSynchronize(g.ev[b]));
        g.pool->copy((uint8_t*)dst + off, g.pin[b], len);
        if (k + Stager::kBufs < nchunks) HIPCHK(c, issue(k + Stager::kBufs));
    }
    g.next = 0;
    return MNW_OK;
}

// An error in the middle of a staged transfer leaves DMAs in flight and the ring's bookkeeping half updated: wait for what was
// enqueued (a synchronous entry point must not return while the destination is still being written, and the next call must not
// refill a pinned buffer whose DMA is still running), then start the ring afresh.  The error that is reported is the first one.
void stager_settle(mnw_ctx* c, hipStream_t s) {
    const std::string first = c->err;
    (void)hipStreamSynchronize(s);
    if (c->stg.ready) { for (int b = 0; b < Stager::kBufs; ++b) c->stg.busy[b] = false; c->stg.next = 0; }
    c->err = first;
}
int copy_file_h2d(mnw_ctx* c, hipStream_t s, void* dev_dst, int fd, int n, const int64_t* file_off, const int64_t* len, const int64_t* dst_off) {
    const int rc = copy_file_h2d_impl(c, s, dev_dst, fd, n, file_off, len, dst_off);
    if (rc) stager_settle(c, s);
    return rc;
}
int copy_d2h_batch(mnw_ctx* c, hipStream_t s, int n, void* const* dst, const void* const* src, const int64_t* len) {
    const int rc = copy_d2h_batch_impl(c, s, n, dst, src, len);
    if (rc) stager_settle(c, s);
    return rc;
}
int copy_d2h(mnw_ctx* c, hipStream_t s, void* dst, const void* dev_src, size_t bytes) {
    const int rc = copy_d2h_impl(c, s, dst, dev_src, bytes);
    if (rc) stager_settle(c, s);
    return rc;
}

// Wait for a stream whose work is expected to finish within microseconds: poll before blocking (a blocking
// hipStreamSynchronize wakes the thread up 10-20 us after the last kernel has finished).
hipError_t wait_stream(hipStream_t s) {
    for (int i = 0; i < 20000; ++i) {
        const hipError_t e = hipStreamQuery(s);
        if (e != hipErrorNotReady) return e;
    }
    return hipStreamSynchronize(s);
}

hipEvent_t get_event(mnw_ctx* c) {
    if (!c->event_pool.empty()) { hipEvent_t e = c->event_pool.back(); c->event_pool.pop_back(); return e; }
    hipEvent_t e = nullptr;
    hipEventCreate(&e);
    return e;
}

struct ScopedTimer {   // event pair around one kernel launch, on the launch's own stream
    mnw_ctx* c; hipStream_t s; TimedLaunch t; bool on;
    ScopedTimer(mnw_ctx* c_, hipStream_t s_, int kernel) : c(c_), s(s_), on(c_->timing) {
        if (on) { t.kernel = kernel; t.start = get_event(c); t.stop = get_event(c); hipEventRecord(t.start, s); }
    }
    ~ScopedTimer() { if (on) { hipEventRecord(t.stop, s); c->launches.push_back(t); } }
};

// the pack pass is LDS-bound at 4 workgroups per CU; more workgroups than that only queue
int pack_grid_for(mnw_ctx* c, int64_t tiles) {
    int cap = c->grid_cap > 0 ? c->grid_cap : c->cus * 4;
    return (int)(tiles < cap ? (tiles > 0 ? tiles : 1) : cap);
}

// Decode launches are NOT persistent: about four tiles (32 KB of float output) per workgroup, tens of thousands of
// workgroups for a long column.  Measured on a 1e9-element column (tools/ab_option.py grid_cap): 2,048 persistent workgroups
// 1.28 ms, 16,384: 1.21, 65,536: 1.15, 131,072: 1.14, 262,144 (two tiles each): 1.29 -- the dispatcher hands out the ranges
// front to back and replaces every workgroup that finishes, which balances better than resident workgroups do (the hardware
// favours a CU's oldest waves), the same way a one-element-per-thread copy (6.26 TB/s) beats every persistent copy (5.5-5.65).
int decode_grid_for(mnw_ctx* c, int64_t tiles) {
    if (c->grid_cap > 0) return (int)(tiles < c->grid_cap ? (tiles > 0 ? tiles : 1) : c->grid_cap);
    int64_t per = tiles / ((int64_t)c->cus * 8);
    per = per < 1 ? 1 : (per > 4 ? 4 : per);
    const int64_t g = (tiles + per - 1) / per;
    return (int)(g < 1 ? 1 : (g > 0x3fffffff ? 0x3fffffff : g));
}

int grid_for(mnw_ctx* c, int64_t tiles) {
    int cap = c->grid_cap > 0 ? c->grid_cap : c->cus * 8;
    return (int)(tiles < cap ? (tiles > 0 ? tiles : 1) : cap);
}

// IEEE float32 quantize step of group.go:318-319 on the host (same arithmetic as the kernels)
float host_quant_floor(float x, float low, float dx) {
    volatile float t = x - low;
    volatile float r = t / dx;
    return floorf(r);
}

uint32_t host_fkey(float f) {
    uint32_t u; memcpy(&u, &f, 4);
    return u ^ ((u >> 31) ? 0xffffffffu : 0x80000000u);
}
float host_funkey(uint32_t k) {
    uint32_t u = k ^ ((k >> 31) ? 0x80000000u : 0xffffffffu);
    float f; memcpy(&f, &u, 4);
    return f;
}

// least float32 x (in the total order -inf .. +inf) whose quantized value is >= h.  q(x) is monotone
// in x for dx > 0, so a bisection over the ordered bit patterns finds it in 32 steps.
float split_point(float low, float dx, int64_t h) {
    uint32_t lo = host_fkey(-INFINITY), hi = host_fkey(INFINITY);   // P(lo) false, P(hi) true
    if ((double)host_quant_floor(-INFINITY, low, dx) >= (double)h) return -INFINITY;
    while (hi - lo > 1) {
        const uint32_t mid = lo + (hi - lo) / 2;
        if ((double)host_quant_floor(host_funkey(mid), low, dx) >= (double)h) hi = mid; else lo = mid;
    }
    return host_funkey(hi);
}

// (high-low)/float32(pixels) in IEEE float32, group.go:316
float quant_dx(float low, float high, int64_t pixels) {
    volatile float span = high - low;
    volatile float p = (float)pixels;
    volatile float dx = span / p;
    return dx;
}

// the reduce pass relies on q(x) being monotone non-decreasing: dx must be positive and finite
bool float_fast_ok(const mnw_enc_job& j) {
    const float dx = quant_dx(j.low, j.high, j.pixels);
    return dx > 0.0f && dx < INFINITY && fabsf(j.low) < INFINITY;
}

bool job_ok(const mnw_enc_job& j) {
    if (j.n < 0 || j.cap < 0) return false;
    if (j.kind != MNW_INT_GROUP && j.kind != MNW_FLOAT_GROUP) return false;
    if (j.n > 0 && (!j.src || (!j.dst && j.cap > 0))) return false;
    if (((uintptr_t)j.dst & 15) != 0) return false;
    if (j.n > (1ll << 47)) return false;
    return true;
}

void fill_enc(DevJob& d, const mnw_enc_job& j) {
    memset(&d, 0, sizeof d);
    d.src = j.src; d.dst = j.dst; d.n = j.n; d.cap = j.cap; d.pixels = j.pixels;
    d.low = j.low; d.high = j.high; d.kind = j.kind; d.periodic = j.periodic;
    d.mode = mnw::kModeStats;
    if (j.kind == MNW_FLOAT_GROUP) {
        // the blocks of one group share (low, high, pixels): derive dx, 1/dx and the split point once
        // per run of equal parameters (the bisection is ~64 float divisions)
        static thread_local struct { float low, high; int64_t pixels; float dxq, xh; double rdx; bool valid; } memo = {0, 0, 0, 0, 0, 0, false};
        if (!memo.valid || memcmp(&memo.low, &j.low, 4) || memcmp(&memo.high, &j.high, 4) || memo.pixels != j.pixels) {
            memo.low = j.low; memo.high = j.high; memo.pixels = j.pixels;
            memo.dxq = quant_dx(j.low, j.high, j.pixels);
            memo.rdx = 1.0 / (double)memo.dxq;
            memo.xh = split_point(j.low, memo.dxq, j.pixels / 2);
            memo.valid = true;
        }
        d.dxq = memo.dxq; d.rdx = memo.rdx; d.xh = memo.xh;
        d.xform = j.xform & (mnw::kXfLog10 | mnw::kXfClamp);
        d.high_below = nextafterf(j.high, -INFINITY);
    }
}

// Launch reduce+pack for device jobs [j0, j1) of one kind, in groups of at most group_bytes of input
// so that the pack pass re-reads its group from the Infinity Cache.  tile0 is group relative.
int launch_encode_range(mnw_ctx* c, JobSlot* sl, hipStream_t s, int j0, int j1, int kind) {
    const int64_t esz = kind == MNW_FLOAT_GROUP ? 4 : 8;
    int g0 = j0;
    while (g0 < j1) {
        int g1 = g0;
        int64_t bytes = 0, tiles = 0;
        while (g1 < j1 && (g1 == g0 || bytes + sl->h_jobs[g1].n * esz <= c->group_bytes)) {
            bytes += sl->h_jobs[g1].n * esz;
            tiles += (sl->h_jobs[g1].n + mnw::kTile - 1) / mnw::kTile;
            ++g1;
        }
        if (tiles > 0) {
            const int grid = grid_for(c, tiles);
            uint64_t* st = sl->d_stats + (size_t)g0 * mnw::kStatSlots * mnw::kStatWords;
            if (kind == MNW_FLOAT_GROUP) {
                void* sy = sync_ptr(sl, g0);
                { ScopedTimer t(c, s, MNW_K_REDUCE_F32); mnw::launch_reduce_f32(s, grid, sl->d_jobs + g0, g1 - g0, tiles, st, sy, sl->d_meta + g0); }
                { ScopedTimer t(c, s, MNW_K_PACK_F32); mnw::launch_pack_f32(s, pack_grid_for(c, tiles), sl->d_jobs + g0, g1 - g0, tiles, sy); }
            } else {
                { ScopedTimer t(c, s, MNW_K_REDUCE_I64); mnw::launch_reduce_i64(s, grid, sl->d_jobs + g0, g1 - g0, tiles, st); }
                { ScopedTimer t(c, s, MNW_K_PACK_I64); mnw::launch_pack_i64(s, grid, sl->d_jobs + g0, g1 - g0, tiles, st, sl->d_meta + g0); }
            }
            HIPCHK(c, hipGetLastError());
        }
        g0 = g1;
    }
    return MNW_OK;
}

// One persistent launch for all float jobs [0, nfloat) of the slot (k_encode_f32).  Groups follow
// the same rule as launch_encode_range; tile0 here is batch-wide, not group relative.
int launch_encode_fused(mnw_ctx* c, JobSlot* sl, hipStream_t s, int nfloat, int nslot_all) {
    if (c->fused_blocks_per_cu == 0) {
        int occ = mnw::encode_f32_blocks_per_cu();
        // the occupancy API can over-report by one block per CU for SGPR-heavy kernels
        // (MI355X_MICROARCH.md "Residency"): never ask for more than 4 (the LDS bound)
        c->fused_blocks_per_cu = occ < 1 ? -1 : (occ > 4 ? 4 : occ);
    }
    if (c->fused_blocks_per_cu < 1) return fail(c, MNW_E_HIP, "occupancy query failed for k_encode_f32");
    int ng = 0;
    int64_t tiles = 0;
    int g0 = 0;
    while (g0 < nfloat) {
        int g1 = g0;
        int64_t bytes = 0;
        sl->h_gtile[ng] = tiles;
        sl->h_gjob[ng] = g0;
        ++ng;
        while (g1 < nfloat && (g1 == g0 || bytes + sl->h_jobs[g1].n * 4 <= c->group_bytes)) {
            bytes += sl->h_jobs[g1].n * 4;
            tiles += (sl->h_jobs[g1].n + mnw::kTile - 1) / mnw::kTile;
            ++g1;
        }
        g0 = g1;
    }
    sl->h_gtile[ng] = tiles;
    sl->h_gjob[ng] = nfloat;
    if (tiles == 0) return MNW_OK;
    int rc = prepare_slot(c, sl, s, nslot_all, true);
    if (rc) return rc;
    int grid = c->cus * c->fused_blocks_per_cu;
    if (c->grid_cap > 0 && c->grid_cap < grid) grid = c->grid_cap;
    { ScopedTimer t(c, s, MNW_K_ENCODE_F32);
      mnw::launch_encode_f32(s, grid, sl->d_jobs, nfloat, sl->d_gtile, sl->d_gjob, ng, sl->d_stats, sync_ptr(sl, 0), abort_ptr(sl), sl->d_meta); }
    HIPCHK(c, hipGetLastError());
    mnw::launch_prepare(s, abort_ptr(sl), sl->d_hsync, 4, nullptr, 0, nullptr, 0);   // abort flag -> pinned host word
    return MNW_OK;
}

// The encode structures other than the single-read kernel do not apply minh's value transform at load time:
// for them the float jobs [0, nfloat) of the slot that carry one are processed into a staging copy first (the
// reference's own data flow: minh.buf, minh.go:128-133) and then encoded from there as plain blocks.
int stage_xforms(mnw_ctx* c, JobSlot* sl, hipStream_t s, int nfloat) {
    size_t total = 0;
    for (int k = 0; k < nfloat; ++k)
        if (sl->h_jobs[k].xform) total += ((size_t)sl->h_jobs[k].n + 3) / 4 * 4;
    if (total == 0) return MNW_OK;
    void* p = c->d_xf;
    int rc = ensure_dev(c, &p, &c->cap_xf, total * sizeof(float));
    c->d_xf = (float*)p;
    if (rc) return rc;
    size_t off = 0;
    mnw::MinhArgs a;
    int m = 0;
    int64_t max_n = 0;
    auto flush = [&]() {
        if (m == 0) return;
        { ScopedTimer t(c, s, MNW_K_MINH_PROCESS); mnw::launch_minh_process(s, c->cus, a, m, max_n); }
        m = 0; max_n = 0;
    };
    memset(&a, 0, sizeof a);
    for (int k = 0; k < nfloat; ++k) {
        DevJob& d = sl->h_jobs[k];
        if (!d.xform) continue;
        if (d.n > 0) {
            mnw::MinhCol& col = a.col[m++];
            col.src = (const float*)d.src; col.dst = c->d_xf + off; col.n = d.n;
            col.low = d.low; col.high = d.high; col.high_below = d.high_below; col.xform = d.xform;
            if (d.n > max_n) max_n = d.n;
            d.src = c->d_xf + off;
            off += ((size_t)d.n + 3) / 4 * 4;
        }
        d.xform = 0;
        if (m == mnw::kMinhBatch) flush();
    }
    flush();
    HIPCHK(c, hipGetLastError());
    return MNW_OK;
}

// Single-read encode: k_encode_reg_f32 for the float jobs [0, nfloat) of the slot and k_encode_reg_i64 for its int jobs
// [nfloat, nslot_all), each only if every block of the kind fits the kernel's sweep (see the kernels' headers).
// *did_float / *did_int say which were launched; the caller sends the other kind through reduce + pack afterwards
// (its prepare_slot uploads the tables again, with the int jobs' item numbering unchanged).
int launch_encode_reg(mnw_ctx* c, JobSlot* sl, hipStream_t s, int nfloat, int nslot_all, bool want_float, bool* did_float, bool* did_int) {
    *did_float = *did_int = false;
    if (c->reg_blocks_per_cu == 0) {
        int occ = mnw::encode_reg_f32_blocks_per_cu();
        c->reg_blocks_per_cu = occ < 1 ? -1 : (occ > 4 ? 4 : occ);     // 256-thread workgroups, 36 KB LDS: four per CU
        occ = mnw::encode_reg_f32_xf_blocks_per_cu();
        c->reg_xf_blocks_per_cu = occ < 1 ? -1 : (occ > 4 ? 4 : occ);  // the instantiation with the value transform (2.5 KB more LDS)
        occ = mnw::encode_reg_i64_blocks_per_cu();
        c->reg_i64_blocks_per_cu = occ < 1 ? -1 : (occ > 4 ? 4 : occ);
    }
    int any_xform = 0;
    for (int k = 0; k < nfloat; ++k) any_xform |= sl->h_jobs[k].xform;
    const int per_cu = any_xform ? c->reg_xf_blocks_per_cu : c->reg_blocks_per_cu;
    // ---- float jobs.  Blocks of at most min(G, 512) items: flat record sweep, weighted hand-out.  Larger ones (`big_blocks` = 0
    // keeps them on two passes): more than `float_flat_items` (256) and at most 2 G' items, G' = G - 8 -- eight sweeper workgroups at
    // the end of the grid derive their descriptors (round 3; 2^22-element blocks 0.55 -> 0.6x of 8 TB/s); up to min(2 G, 4096) items
    // (2^24 elements on a full grid of G = 1,024, which leaves no workgroup over): the two-level sweep.  Beyond G' (G) items the
    // rounds must be unweighted (see "large blocks" in mnw_kernels.hip).
    const int64_t item = mnw::reg_item_elems();
    int grid = c->cus * per_cu;
    if (c->grid_cap > 0 && c->grid_cap < grid) grid = c->grid_cap;
    int64_t items = 0, groups = 0, max_ni = 0;
    int nsw_f = 0, flat_f = (int)mnw::reg_max_items_per_block();
    bool fl = want_float && nfloat > 0 && per_cu >= 1;
    if (fl) {
        for (int k = 0; k < nfloat; ++k) {
            const int64_t ni = (sl->h_jobs[k].n + item - 1) / item;
            if (ni > max_ni) max_ni = ni;
            items += ni;
        }
        const int full = grid;
        if (items < grid) grid = items > 0 ? (int)items : 1;
        const int64_t flat = mnw::reg_max_items_per_block();
        int64_t lim = grid < flat ? grid : flat;
        if (c->big_blocks) {
            const int nsw = mnw::encode_reg_f32_sweepers();
            const int gi = full - nsw < items ? full - nsw : (int)items;          // item workgroups if the sweepers come out of the full grid
            const int64_t fsw = c->float_flat_items < flat ? c->float_flat_items : flat;
            if (max_ni > (gi < fsw ? gi : fsw) && gi >= 1 && max_ni <= 2 * (int64_t)gi) {
                nsw_f = nsw; flat_f = (int)(gi < fsw ? gi : fsw); grid = gi + nsw; lim = 2 * (int64_t)gi;
            } else {
                lim = 2 * (int64_t)grid;
                if (lim > mnw::reg_max_items_per_big_block()) lim = mnw::reg_max_items_per_big_block();
            }
        }
        if (max_ni > lim) fl = false;
    }
    if (!fl) items = 0;
    // ---- int jobs.  Blocks of at most min(G, 512) items: flat sweep.  Larger ones, up to 2 G items (2^23 elements on a full grid):
    // sweeper workgroups at the end of the grid derive their descriptors, rounds unweighted (k_encode_reg_i64's header).
    const int64_t item_i = mnw::reg_item_elems_i64();
    int grid_i = c->cus * c->reg_i64_blocks_per_cu;
    if (c->grid_cap > 0 && c->grid_cap < grid_i) grid_i = c->grid_cap;
    int64_t items_i = 0;
    int nsw_i = 0, flat_i = 0;
    bool in = c->single_read_int && nslot_all > nfloat && c->reg_i64_blocks_per_cu >= 1;
    if (in) {
        int64_t mx = 0;
        for (int k = nfloat; k < nslot_all; ++k) {
            const int64_t ni = (sl->h_jobs[k].n + item_i - 1) / item_i;
            if (ni > mx) mx = ni;
            items_i += ni;
        }
        const int full = grid_i;
        if (items_i < grid_i) grid_i = items_i > 0 ? (int)items_i : 1;
        const int64_t flat = c->big_blocks && c->int_flat_items < mnw::reg_max_items_per_block() ? c->int_flat_items : mnw::reg_max_items_per_block();
        flat_i = (int)(grid_i < flat ? grid_i : flat);
        if (mx > flat_i) {
            // item workgroups + sweepers must all be resident: the sweepers come out of the full grid
            nsw_i = mnw::encode_reg_i64_sweepers();
            const int gi = full - nsw_i < items_i ? full - nsw_i : (int)items_i;
            flat_i = (int)(gi < flat ? gi : flat);
            if (!c->big_blocks || gi < 1 || mx > 2 * (int64_t)gi) { in = false; nsw_i = 0; }
            else grid_i = gi + nsw_i;
        }
    }
    if (!in) items_i = 0;
    if (!fl && !in) return MNW_OK;
    const bool unweighted = max_ni > grid - nsw_f;
    if (fl) {
        items = 0;
        for (int k = 0; k < nfloat; ++k) {
            const int64_t ni = (sl->h_jobs[k].n + item - 1) / item;
            sl->h_jobs[k].tile0 = items;          // first ITEM of the job in this launch
            sl->h_jobs[k].grp0 = groups;          // ... and its group records: R x R of them (R copies of R records), large blocks only
            items += ni;
            if (nsw_f == 0 && ni > mnw::reg_max_items_per_block()) {
                const int64_t R = (ni + mnw::reg_group_items() - 1) / mnw::reg_group_items();
                groups += R * R;
            }
        }
    }
    const int64_t items_pad = (items + 3) / 4 * 4;       // the int launch's item table starts here
    if (in) {
        int64_t t = 0;
        for (int k = nfloat; k < nslot_all; ++k) { sl->h_jobs[k].tile0 = t; t += (sl->h_jobs[k].n + item_i - 1) / item_i; }
    }
    const int64_t fdrecs = nsw_f ? nfloat : 0;            // one descriptor record per float job, behind the group records
    const size_t need = (size_t)(items + groups + fdrecs + items_i + (nsw_i ? nslot_all - nfloat : 0)) * mnw::item_rec_bytes();      // (+ one descriptor record per int job)
    if (need > c->cap_recs) {
        // (re)allocation happens between batches only: the previous batch has been finished by then
        if (c->d_recs) { HIPCHK(c, hipStreamSynchronize(s)); HIPCHK(c, hipFree(c->d_recs)); }
        c->d_recs = nullptr; c->cap_recs = 0;
        const size_t want = need + need / 2 + 4096;
        HIPCHK(c, hipMalloc(&c->d_recs, want));
        HIPCHK(c, hipMemsetAsync(c->d_recs, 0, want, s));      // tags of a fresh buffer must not match any generation
        c->cap_recs = want;
    }
    const int64_t table_items = items_pad + items_i;
    if ((size_t)table_items > sl->cap_itemjob) {
        // the slot is not in flight (acquire_slot waited for its previous batch)
        if (sl->d_itemjob) { hipFree(sl->d_itemjob); hipHostFree(sl->h_itemjob); }
        sl->d_itemjob = nullptr; sl->h_itemjob = nullptr; sl->cap_itemjob = 0;
        const size_t want = ((size_t)table_items + (size_t)table_items / 2 + 1024 + 3) / 4 * 4;
        HIPCHK(c, hipMalloc((void**)&sl->d_itemjob, want * sizeof(int)));
        HIPCHK(c, hipHostMalloc((void**)&sl->h_itemjob, want * sizeof(int), hipHostMallocCoherent | hipHostMallocMapped));
        sl->cap_itemjob = want;
        sl->shadow_tag = 0;                   // a fresh device table: nothing to reuse
    }
    c->reg_gen = (c->reg_gen + 1) & 0x7fffffffu;
    if (c->reg_gen == 0) c->reg_gen = 1;
    sl->h_sync[0] = 0;                        // the kernels raise it (pinned host word) if they give up
    // (jobs of the batch that do not go through these kernels take the two-kernel path afterwards and need their
    // statistics and synchronisation words zeroed; the jobs here need neither)
    const bool others = (nfloat > 0 && !fl) || (nslot_all > nfloat && !in);
    const size_t table_bytes = sizeof(DevJob) * (size_t)nslot_all;
    const int tag = in ? (fl ? 4 : 3) : 2;
    if (others || !tables_unchanged(sl, tag, table_bytes)) {
        if (fl)
            for (int k = 0; k < nfloat; ++k) {
                const int64_t i0 = sl->h_jobs[k].tile0, ni = (sl->h_jobs[k].n + item - 1) / item;
                for (int64_t i = 0; i < ni; ++i) sl->h_itemjob[i0 + i] = k;
            }
        for (int64_t i = items; i < items_pad; ++i) sl->h_itemjob[i] = 0;
        if (in)
            for (int k = nfloat; k < nslot_all; ++k) {
                const int64_t i0 = items_pad + sl->h_jobs[k].tile0, ni = (sl->h_jobs[k].n + item_i - 1) / item_i;
                for (int64_t i = 0; i < ni; ++i) sl->h_itemjob[i0 + i] = k - nfloat;     // relative to the int launch's job table
            }
        // one small launch: job records and item table to device memory, abort word zeroed
        mnw::launch_prepare(s, sl->h_up, sl->d_up, (int64_t)sl->up_bytes,
                            others ? sl->d_stats : nullptr, others ? (int64_t)(sizeof(uint64_t) * mnw::kStatSlots * mnw::kStatWords * nslot_all) : 0,
                            sl->d_sync, (int64_t)(sizeof(uint32_t) * kSyncPad + (others ? sl->sync_stride * nslot_all : 0)),
                            sl->h_itemjob, sl->d_itemjob, ((int64_t)table_items * 4 + 15) / 16 * 16);
        HIPCHK(c, hipGetLastError());
        if (others) sl->shadow_tag = 0; else remember_tables(sl, tag, table_bytes);
    }
    if (fl && items > 0) {
        ScopedTimer t(c, s, MNW_K_ENCODE_REG_F32);
        // the weighted hand-out only when the grid is the full co-resident one (four workgroups on every CU) and no
        // block has more items than the grid has workgroups; group records live behind the item records
        mnw::launch_encode_reg_f32(s, grid, sl->d_jobs, nfloat, items, sl->d_itemjob, c->d_recs,
                                   (uint8_t*)c->d_recs + (size_t)items * mnw::item_rec_bytes(), c->reg_gen, abort_ptr(sl), sl->d_hsync,
                                   sl->d_meta, (grid == c->cus * 4 && !unweighted && c->reg_weighted) ? 1 + c->reg_round_table : 0, c->reg_drop_item, any_xform,
                                   max_ni > flat_f ? 1 : 0, (uint8_t*)c->d_recs + (size_t)(items + groups) * mnw::item_rec_bytes(), nsw_f, flat_f);
    }
    if (in && items_i > 0) {
        ScopedTimer t(c, s, MNW_K_ENCODE_REG_I64);
        mnw::launch_encode_reg_i64(s, grid_i, sl->d_jobs + nfloat, nslot_all - nfloat, items_i, sl->d_itemjob + items_pad,
                                   (uint8_t*)c->d_recs + (size_t)(items + groups + fdrecs) * mnw::item_rec_bytes(), c->reg_gen, abort_ptr(sl),
                                   sl->d_hsync, sl->d_meta + nfloat, (grid_i == c->cus * 4 && c->reg_weighted && nsw_i == 0) ? 1 + c->reg_round_table : 0, c->reg_drop_item,
                                   (uint8_t*)c->d_recs + (size_t)(items + groups + fdrecs + items_i) * mnw::item_rec_bytes(), nsw_i, flat_i);
    }
    HIPCHK(c, hipGetLastError());
    *did_float = fl; *did_int = in;
    return MNW_OK;
}

// Encode int64 blocks that already sit in device memory, synchronously (used by the exact path and
// by the host-pointer entry points).  Uses device job slot 0.
int encode_one_sync(mnw_ctx* c, hipStream_t s, const mnw_enc_job& job, int mode, int64_t emin, int ebits, Meta* out) {
    JobSlot* sl;
    int rc = acquire_slot(c, 1, &sl);
    if (rc) return rc;
    DevJob& d = sl->h_jobs[0];
    fill_enc(d, job);
    d.mode = mode; d.min = emin; d.bits = ebits; d.tile0 = 0;
    rc = prepare_slot(c, sl, s, 1, true);
    if (rc) return rc;
    const int64_t tiles = (job.n + mnw::kTile - 1) / mnw::kTile;
    const int grid = grid_for(c, tiles);
    if (job.kind == MNW_FLOAT_GROUP) {
        { ScopedTimer t(c, s, MNW_K_REDUCE_F32); mnw::launch_reduce_f32(s, grid, sl->d_jobs, 1, tiles, sl->d_stats, sync_ptr(sl, 0), sl->d_meta); }
        { ScopedTimer t(c, s, MNW_K_PACK_F32); mnw::launch_pack_f32(s, pack_grid_for(c, tiles), sl->d_jobs, 1, tiles, sync_ptr(sl, 0)); }
    } else {
        if (mode == mnw::kModeStats) { ScopedTimer t(c, s, MNW_K_REDUCE_I64); mnw::launch_reduce_i64(s, grid, sl->d_jobs, 1, tiles, sl->d_stats); }
        { ScopedTimer t(c, s, MNW_K_PACK_I64); mnw::launch_pack_i64(s, grid, sl->d_jobs, 1, tiles, sl->d_stats, sl->d_meta); }
    }
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, hipStreamSynchronize(s));
    *out = sl->h_meta[0];
    return MNW_OK;
}

// floatGroup.writeData for a block whose periodicMin depends on element order: the literal sequence
// of group.go:312-327 on the GPU (quantize -> sequential periodicMin -> bound -> intGroup path).
int encode_exact_float(mnw_ctx* c, hipStream_t s, const mnw_enc_job& job_in, Meta* out) {
    mnw_enc_job job = job_in;
    if (job.xform & (mnw::kXfLog10 | mnw::kXfClamp)) {
        // minh's value transform first (all kernels of the batch have finished: the staging copy is free)
        void* p = c->d_xf;
        int rx = ensure_dev(c, &p, &c->cap_xf, ((size_t)job.n + 4) * sizeof(float));
        c->d_xf = (float*)p;
        if (rx) return rx;
        mnw::MinhArgs a;
        memset(&a, 0, sizeof a);
        a.col[0].src = (const float*)job.src; a.col[0].dst = c->d_xf; a.col[0].n = job.n;
        a.col[0].low = job.low; a.col[0].high = job.high; a.col[0].high_below = nextafterf(job.high, -INFINITY);
        a.col[0].xform = job.xform & (mnw::kXfLog10 | mnw::kXfClamp);
        { ScopedTimer t(c, s, MNW_K_MINH_PROCESS); mnw::launch_minh_process(s, c->cus, a, 1, job.n); }
        job.src = c->d_xf;
        job.xform = 0;
    }
    int rc = ensure_dev(c, (void**)&c->d_q, &c->cap_q, (size_t)job.n * 8);
    if (rc) return rc;
    if (!c->d_state) HIPCHK(c, hipMalloc((void**)&c->d_state, 4 * sizeof(int64_t)));
    const float dx = quant_dx(job.low, job.high, job.pixels);
    const int grid = (int)((job.n + mnw::kThreads - 1) / mnw::kThreads < c->cus * 8 ? (job.n + mnw::kThreads - 1) / mnw::kThreads : c->cus * 8);
    mnw::launch_quantize_i64(s, grid, (const float*)job.src, job.n, job.low, dx, c->d_q);
    if (job.periodic == 1) {
        int64_t st[4] = {0, 1, 0, 0};
        HIPCHK(c, hipMemcpyAsync(c->d_state, st, sizeof st, hipMemcpyHostToDevice, s));
        HIPCHK(c, hipStreamSynchronize(s));
        const int64_t budget = 1ll << 22;
        for (int64_t done = 0; done < job.n; done += budget)
            mnw::launch_periodic_min_seq(s, c->d_q, job.n, job.pixels, budget, c->d_state);
        HIPCHK(c, hipMemcpyAsync(st, c->d_state, sizeof st, hipMemcpyDeviceToHost, s));
        HIPCHK(c, hipStreamSynchronize(s));
        const int64_t pmin = st[2] ? 0 : st[0];
        mnw::launch_bound_i64(s, grid, c->d_q, job.n, pmin, job.pixels);
    }
    HIPCHK(c, hipGetLastError());
    mnw_enc_job ij = job;
    ij.kind = MNW_INT_GROUP;
    ij.src = c->d_q;
    return encode_one_sync(c, s, ij, mnw::kModeStats, 0, 0, out);
}

int status_of(const Meta& m) { return (int)m.status; }

}  // namespace

// ---------------------------------------------------------------------------------------------
extern "C" {

int mnw_abi_version(void) { return MNW_ABI_VERSION; }

int mnw_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

const char* mnw_strerror(int status) {
    switch (status) {
        case MNW_OK: return "ok";
        case MNW_E_ARG: return "bad argument (null pointer, negative length, bits > 64, misaligned dst)";
        case MNW_E_BUFLEN: return "buffer length does not match ArrayBytes(bits, n)";
        case MNW_E_NOGPU: return "no HIP device available (libminnow_hip has no CPU fallback)";
        case MNW_E_HIP: return "HIP runtime error";
        case MNW_E_CAPACITY: return "packed stream does not fit the output capacity";
        case MNW_E_DOMAIN: return "bit width outside 0..64 (max-min == 2^64-1)";
        case MNW_E_NOMEM: return "out of memory";
        case MNW_E_IO: return "file I/O error (read failed or the file is shorter than the ranges asked for)";
        default: return "unknown status";
    }
}

const char* mnw_last_error(const mnw_ctx* ctx) { return ctx ? ctx->err.c_str() : "null context"; }

const char* mnw_kernel_name(int k) {
    switch (k) {
        case MNW_K_REDUCE_F32: return "k_reduce_f32";
        case MNW_K_PACK_F32: return "k_pack_f32";
        case MNW_K_REDUCE_I64: return "k_reduce_i64";
        case MNW_K_PACK_I64: return "k_pack_i64";
        case MNW_K_UNPACK_F32: return "k_unpack_f32";
        case MNW_K_UNPACK_I64: return "k_unpack_i64";
        case MNW_K_ENCODE_F32: return "k_encode_f32";
        case MNW_K_ENCODE_REG_F32: return "k_encode_reg_f32";
        case MNW_K_ENCODE_REG_I64: return "k_encode_reg_i64";
        case MNW_K_ENCODE_REG_VEC3: return "k_encode_reg_vec3";
        case MNW_K_SUBCELL_GATHER: return "k_subcell_gather";
        case MNW_K_SUBCELL_SCATTER: return "k_subcell_scatter";
        case MNW_K_BOUNDS3: return "k_bounds3";
        case MNW_K_REDUCE_VEC3: return "k_reduce_vec3";
        case MNW_K_PACK_VEC3: return "k_pack_vec3";
        case MNW_K_UNPACK_VEC3: return "k_unpack_vec3";
        case MNW_K_MINH_PROCESS: return "k_minh_process";
        case MNW_K_MINH_UNLOG: return "k_minh_unlog";
        case MNW_K_TEXT_LINES: return "k_text_count+k_text_scan";
        case MNW_K_TEXT_PARSE: return "k_text_parse";
        case MNW_K_CUT: return "k_cut_count+scan / k_cut_scatter";
        default: return "";
    }
}

int mnw_ctx_create(int device, mnw_ctx** out) {
    if (!out) return MNW_E_ARG;
    *out = nullptr;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) return MNW_E_NOGPU;
    if (device < 0 || device >= n) return MNW_E_ARG;
    mnw_ctx* c = new (std::nothrow) mnw_ctx();
    if (!c) return MNW_E_NOMEM;
    c->device = device;
    memset(c->acc, 0, sizeof c->acc);
    if (hipSetDevice(device) != hipSuccess) { delete c; return MNW_E_HIP; }
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) == hipSuccess) c->cus = prop.multiProcessorCount;
    if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess) { delete c; return MNW_E_HIP; }
    *out = c;
    return MNW_OK;
}

int mnw_ctx_destroy(mnw_ctx* c) {
    if (!c) return MNW_E_ARG;
    hipSetDevice(c->device);
    hipDeviceSynchronize();
    for (auto& t : c->launches) { hipEventDestroy(t.start); hipEventDestroy(t.stop); }
    for (auto e : c->event_pool) hipEventDestroy(e);
    for (auto& sl : c->slots) { free_slot(sl); if (sl.consumed) hipEventDestroy(sl.consumed); }
    for (auto& sl : c->dslots) { free_slot(sl); if (sl.consumed) hipEventDestroy(sl.consumed); }
    stager_free(c);
    if (c->d_in) hipFree(c->d_in);
    if (c->d_out) hipFree(c->d_out);
    if (c->d_q) hipFree(c->d_q);
    if (c->d_state) hipFree(c->d_state);
    if (c->d_recs) hipFree(c->d_recs);
    if (c->d_soa) hipFree(c->d_soa);
    if (c->d_xf) hipFree(c->d_xf);
    for (void* q : {c->d_text, c->d_tile, c->d_tslow, c->d_tstate}) if (q) hipFree(q);
    if (c->d_bkeys) hipFree(c->d_bkeys);
    if (c->h_bkeys) hipHostFree(c->h_bkeys);
    if (c->stream) hipStreamDestroy(c->stream);
    delete c;
    return MNW_OK;
}

int mnw_set_option(mnw_ctx* c, const char* name, int64_t value) {
    if (!c || !name) return MNW_E_ARG;
    if (!strcmp(name, "group_bytes")) { c->group_bytes = value > 0 ? value : (1ll << 40); return MNW_OK; }
    if (!strcmp(name, "fused")) { c->fused = value != 0; return MNW_OK; }
    if (!strcmp(name, "single_read")) { c->single_read = value != 0; return MNW_OK; }
    if (!strcmp(name, "single_read_int")) { c->single_read_int = value != 0; return MNW_OK; }
    if (!strcmp(name, "vec3_single_read")) { c->vec3_single_read = value != 0; return MNW_OK; }
    if (!strcmp(name, "unlog_fused")) { c->unlog_fused = value != 0; return MNW_OK; }
    if (!strcmp(name, "single_read_backoff")) { c->single_read_backoff = value != 0; c->sr_backoff = c->sr_backoff_left = 0; return MNW_OK; }
    if (!strcmp(name, "reg_round_table")) { if (value < 0 || value > 4) return fail(c, MNW_E_ARG, "reg_round_table: 0..4"); c->reg_round_table = (int)value; return MNW_OK; }
    if (!strcmp(name, "big_blocks")) { c->big_blocks = value != 0; return MNW_OK; }
    if (!strcmp(name, "vec3_tiles_per_wg")) { c->vec3_tiles_per_wg = (int)value; return MNW_OK; }
    if (!strcmp(name, "host_threads")) { if (c->stg.ready) return fail(c, MNW_E_ARG, "host_threads must be set before the first host-pointer call"); c->host_threads = (int)value; return MNW_OK; }
    if (!strcmp(name, "stage_min_bytes")) { c->stage_min = value; return MNW_OK; }
    if (!strcmp(name, "vec3_fused")) { c->vec3_fused = value != 0; return MNW_OK; }
    if (!strcmp(name, "vec3_deep")) { c->vec3_deep = value < 3 ? 3 : value > 5 ? 5 : (int)value; c->reg_vec3_blocks_per_cu = 0; return MNW_OK; }
    if (!strcmp(name, "vec3_round_table")) { c->vec3_round_table = (int)value; return MNW_OK; }
    if (!strcmp(name, "float_flat_items")) { if (value < 1) return MNW_E_ARG; c->float_flat_items = (int)(value > 512 ? 512 : value); return MNW_OK; }
    if (!strcmp(name, "int_flat_items")) { if (value < 1) return MNW_E_ARG; c->int_flat_items = (int)(value > 512 ? 512 : value); return MNW_OK; }
    if (!strcmp(name, "vec3_sweepers")) { mnw::vec3_reg_set_sweepers((int)value); return MNW_OK; }      // (process-wide)
    if (!strcmp(name, "vec3_unpack_tiles")) { if (value < 1 || value > 1024) return MNW_E_ARG; c->vec3_unpack_tiles = (int)value; return MNW_OK; }
    if (!strcmp(name, "vec3_x_order")) { if (value < 0 || value > 2) return MNW_E_ARG; c->vec3_x_order = (int)value; return MNW_OK; }
    if (!strcmp(name, "vec3_unpack_db")) { c->vec3_unpack_db = value != 0; return MNW_OK; }
    if (!strcmp(name, "reg_weighted")) { c->reg_weighted = value != 0; return MNW_OK; }
    if (!strcmp(name, "reg_test_drop_item")) { c->reg_drop_item = value; return MNW_OK; }
    if (!strcmp(name, "grid_cap")) { c->grid_cap = (int)value; return MNW_OK; }
    return fail(c, MNW_E_ARG, "unknown option");
}

// ---- scalars ------------------------------------------------------------------------------
int64_t mnw_precision_needed(uint64_t max) {
    int b = mnw::precision_needed(max);
    return b < 0 ? INT64_MIN : b;            // int(-Inf) on amd64 (bit.go:20 with max+1 == 0)
}

int64_t mnw_array_bytes(int64_t bits, int64_t length) { return mnw::array_bytes(bits, length); }

int64_t mnw_float_group_pixels(float low, float high, float dx) {   // writer.go:73
    volatile float span = high - low;
    volatile float r = span / dx;
    double c = ceil((double)r);
    if (!(c >= -9223372036854775808.0 && c < 9223372036854775808.0)) return INT64_MIN;
    return (int64_t)c;
}

// ---- timing -------------------------------------------------------------------------------
int mnw_timing_enable(mnw_ctx* c, int on) { if (!c) return MNW_E_ARG; c->timing = on != 0; return MNW_OK; }

static int collect_timing(mnw_ctx* c) {
    for (auto& t : c->launches) {
        float ms = 0;
        hipEventSynchronize(t.stop);
        if (hipEventElapsedTime(&ms, t.start, t.stop) == hipSuccess) {
            c->acc[t.kernel].total_ms += ms;
            c->acc[t.kernel].launches += 1;
        }
        c->event_pool.push_back(t.start);
        c->event_pool.push_back(t.stop);
    }
    c->launches.clear();
    return MNW_OK;
}

int mnw_timing_reset(mnw_ctx* c) {
    if (!c) return MNW_E_ARG;
    collect_timing(c);
    memset(c->acc, 0, sizeof c->acc);
    return MNW_OK;
}

int mnw_timing_get(mnw_ctx* c, mnw_kernel_time* out) {
    if (!c || !out) return MNW_E_ARG;
    collect_timing(c);
    memcpy(out, c->acc, sizeof c->acc);
    return MNW_OK;
}

int mnw_counters_get(mnw_ctx* c, int64_t out[MNW_C_COUNT]) {
    if (!c || !out) return MNW_E_ARG;
    memcpy(out, c->counters, sizeof c->counters);
    return MNW_OK;
}
int mnw_counters_reset(mnw_ctx* c) {
    if (!c) return MNW_E_ARG;
    memset(c->counters, 0, sizeof c->counters);
    return MNW_OK;
}

// ---- batched device-pointer encode ----------------------------------------------------------
int mnw_encode_blocks_async(mnw_ctx* c, mnw_stream_t stream, const mnw_enc_job* jobs, int njobs) {
    if (!c || njobs < 0 || (njobs > 0 && !jobs)) return fail(c, MNW_E_ARG, "encode: bad arguments");
    if (c->pending) return fail(c, MNW_E_ARG, "encode: a batch is already pending; call mnw_encode_blocks_finish");
    HIPCHK(c, hipSetDevice(c->device));
    hipStream_t s = stream ? (hipStream_t)stream : c->stream;
    for (int i = 0; i < njobs; ++i)
        if (!job_ok(jobs[i])) return fail(c, MNW_E_ARG, "encode: bad job (null/misaligned pointer, negative size or unknown kind)");
    JobSlot* sl;
    int rc = acquire_slot(c, njobs > 0 ? njobs : 1, &sl);
    if (rc) return rc;

    c->pend_jobs.assign(jobs, jobs + njobs);
    c->pend_slot.assign(njobs, -1);
    c->pend_host.assign(njobs, Meta{0, 0, 0, 0});
    c->pend_stream = s;

    // device slots: float jobs first, then int jobs, each in caller order
    int nslot = 0, nfloat = 0;
    for (int pass = 0; pass < 2; ++pass) {
        const int kind = pass == 0 ? MNW_FLOAT_GROUP : MNW_INT_GROUP;
        for (int i = 0; i < njobs; ++i) {
            const mnw_enc_job& j = jobs[i];
            if (j.kind != kind) continue;
            if (j.n == 0) continue;   // group.go:351, bit.go:152: empty block -> min 0, bits 0, no bytes
            if (kind == MNW_FLOAT_GROUP && (j.pixels > 0x7fffffffll || j.pixels <= 0 || !float_fast_ok(j))) {
                c->pend_host[i].status = mnw::kNeedsExact;   // 64-bit ring: exact path at finish
                continue;
            }
            fill_enc(sl->h_jobs[nslot], j);
            c->pend_slot[i] = nslot++;
        }
        if (pass == 0) nfloat = nslot;
    }
    if (nslot > 0) {
        // group-relative tile numbering is assigned inside launch_encode_range's grouping; compute it here
        for (int pass = 0; pass < 2; ++pass) {
            const int j0 = pass == 0 ? 0 : nfloat, j1 = pass == 0 ? nfloat : nslot;
            const int64_t esz = pass == 0 ? 4 : 8;
            int g0 = j0;
            while (g0 < j1) {
                int g1 = g0; int64_t bytes = 0, tiles = 0;
                while (g1 < j1 && (g1 == g0 || bytes + sl->h_jobs[g1].n * esz <= c->group_bytes)) {
                    sl->h_jobs[g1].tile0 = tiles;
                    bytes += sl->h_jobs[g1].n * esz;
                    tiles += (sl->h_jobs[g1].n + mnw::kTile - 1) / mnw::kTile;
                    ++g1;
                }
                g0 = g1;
            }
        }
        const bool fused = c->fused && nfloat > 0;
        if (fused) {
            // the fused launch numbers the tiles of the float jobs batch-wide (the int jobs keep the
            // group-relative numbering assigned above)
            int64_t tiles = 0;
            for (int k = 0; k < nfloat; ++k) { sl->h_jobs[k].tile0 = tiles; tiles += (sl->h_jobs[k].n + mnw::kTile - 1) / mnw::kTile; }
        }
        c->pend_fused = fused;
        c->pend_nfloat = nfloat;
        c->pend_mode = fused ? 1 : 0;
        c->pend_int_reg = false;
        c->pend_nslot = nslot;
        bool did_f = false, did_i = false;
        // Something else is running on this GPU if the last attempt gave up: do not pay its 50 ms bound again at once
        bool backing_off = false;
        if (!fused && c->single_read_backoff && c->sr_backoff_left > 0 && ((c->single_read && nfloat > 0) || (c->single_read_int && nslot > nfloat))) {
            --c->sr_backoff_left;
            c->counters[MNW_C_BACKOFF_BATCHES]++;
            backing_off = true;
        }
        if (!fused && !backing_off && ((c->single_read && nfloat > 0) || (c->single_read_int && nslot > nfloat))) {
            // keep the group-relative numbering of what turns out not to be eligible
            std::vector<int64_t> keep(nslot);
            for (int k = 0; k < nslot; ++k) keep[k] = sl->h_jobs[k].tile0;
            rc = launch_encode_reg(c, sl, s, nfloat, nslot, c->single_read != 0, &did_f, &did_i);
            if (rc) return rc;
            if (!did_f) for (int k = 0; k < nfloat; ++k) sl->h_jobs[k].tile0 = keep[k];
            if (!did_i) for (int k = nfloat; k < nslot; ++k) sl->h_jobs[k].tile0 = keep[k];
            if (did_f) { c->pend_mode = 2; c->counters[MNW_C_SINGLE_READ_BATCHES]++; }
            if (did_i) { c->pend_int_reg = true; c->counters[MNW_C_INT_SINGLE_READ_BATCHES]++; }
        }
        if (!did_f) {
            // not the single-read kernel: minh value transforms are applied in a pass of their own first
            int rs = stage_xforms(c, sl, s, nfloat);
            if (rs) return rs;
        }
        rc = MNW_OK;
        if (did_f) {
            // launched above
        } else if (fused) {
            c->counters[MNW_C_FUSED_BATCHES]++;
            rc = launch_encode_fused(c, sl, s, nfloat, nslot);        // its k_prepare covers all nslot jobs
        } else if (nfloat > 0 || !did_i) {
            if (nfloat > 0) c->counters[MNW_C_TWO_PASS_BATCHES]++;
            rc = prepare_slot(c, sl, s, nslot, true);
            if (rc) return rc;
            rc = launch_encode_range(c, sl, s, 0, nfloat, MNW_FLOAT_GROUP);
        }
        if (rc) return rc;
        if (!did_i) {
            rc = launch_encode_range(c, sl, s, nfloat, nslot, MNW_INT_GROUP);
            if (rc) return rc;
        }
    }
    rc = release_slot(c, sl, s);
    if (rc) return rc;
    c->pending = true;
    c->pend = sl;
    return MNW_OK;
}

int mnw_encode_blocks_finish(mnw_ctx* c, mnw_stream_t stream, mnw_block_meta* meta) {
    if (!c || !c->pending) return fail(c, MNW_E_ARG, "finish: no pending batch");
    (void)stream;
    hipStream_t s = c->pend_stream;
    c->pending = false;
    JobSlot* sl = c->pend;
    c->pend = nullptr;
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, wait_stream(s));
    const int njobs = (int)c->pend_jobs.size();
    const bool f_reg = c->pend_mode != 0 && c->pend_nfloat > 0, i_reg = c->pend_int_reg;
    c->pend_int_reg = false;
    if ((f_reg || i_reg) && sl->h_sync[0] != 0) {
        // The persistent launch gave up (its workgroups were not all co-resident, e.g. another kernel held
        // part of the GPU): rerun its jobs through the two-kernel path.  Still the GPU path.
        c->counters[MNW_C_SINGLE_READ_ABORTS]++;
        c->sr_backoff = c->sr_backoff ? (c->sr_backoff < 64 ? 2 * c->sr_backoff : 64) : 1;
        c->sr_backoff_left = c->sr_backoff;
        sl->shadow_tag = 0;                       // the abort word on the device is set: the next batch must prepare
        for (int pass = 0; pass < 2; ++pass) {    // group-relative tile numbering again
            if (!(pass == 0 ? f_reg : i_reg)) continue;
            const int j0 = pass == 0 ? 0 : c->pend_nfloat, j1 = pass == 0 ? c->pend_nfloat : c->pend_nslot;
            const int64_t esz = pass == 0 ? 4 : 8;
            for (int g0 = j0; g0 < j1; ) {
                int g1 = g0; int64_t bytes = 0, tiles = 0;
                while (g1 < j1 && (g1 == g0 || bytes + sl->h_jobs[g1].n * esz <= c->group_bytes)) {
                    sl->h_jobs[g1].tile0 = tiles;
                    bytes += sl->h_jobs[g1].n * esz;
                    tiles += (sl->h_jobs[g1].n + mnw::kTile - 1) / mnw::kTile;
                    ++g1;
                }
                g0 = g1;
            }
        }
        int rc = f_reg ? stage_xforms(c, sl, s, c->pend_nfloat) : MNW_OK;
        if (rc) return rc;
        rc = prepare_slot(c, sl, s, c->pend_nslot, true);
        if (rc) return rc;
        if (f_reg) rc = launch_encode_range(c, sl, s, 0, c->pend_nfloat, MNW_FLOAT_GROUP);
        if (rc) return rc;
        if (i_reg) rc = launch_encode_range(c, sl, s, c->pend_nfloat, c->pend_nslot, MNW_INT_GROUP);
        if (rc) return rc;
        HIPCHK(c, hipStreamSynchronize(s));
    } else if (f_reg || i_reg) {
        c->sr_backoff = 0;                        // a single-read launch ran to its end: nothing in the way any more
    }
    std::vector<Meta> res(njobs);
    for (int i = 0; i < njobs; ++i) res[i] = c->pend_slot[i] >= 0 ? sl->h_meta[c->pend_slot[i]] : c->pend_host[i];
    int first = MNW_OK;
    for (int i = 0; i < njobs; ++i) {
        if (res[i].status == mnw::kNeedsExact) {
            Meta m;
            c->counters[MNW_C_EXACT_BLOCKS]++;
            int rc = encode_exact_float(c, s, c->pend_jobs[i], &m);
            if (rc) return rc;
            res[i] = m;
        }
        if (res[i].status != 0 && first == MNW_OK) first = (int)res[i].status;
    }
    if (meta) memcpy(meta, res.data(), sizeof(Meta) * njobs);
    if (first != MNW_OK) fail(c, first, mnw_strerror(first));
    return first;
}

int mnw_encode_blocks(mnw_ctx* c, mnw_stream_t stream, const mnw_enc_job* jobs, int njobs, mnw_block_meta* meta) {
    int rc = mnw_encode_blocks_async(c, stream, jobs, njobs);
    if (rc) return rc;
    return mnw_encode_blocks_finish(c, stream, meta);
}

// ---- batched device-pointer decode ----------------------------------------------------------
int mnw_decode_blocks(mnw_ctx* c, mnw_stream_t stream, const mnw_dec_job* jobs, int njobs) {
    if (!c || njobs < 0 || (njobs > 0 && !jobs)) return fail(c, MNW_E_ARG, "decode: bad arguments");
    if (c->pending) return fail(c, MNW_E_ARG, "decode: an encode batch is pending");
    HIPCHK(c, hipSetDevice(c->device));
    hipStream_t s = stream ? (hipStream_t)stream : c->stream;
    // every job is checked before a ring slot is taken: a rejected batch leaves the context as it was
    for (int i = 0; i < njobs; ++i) {
        const mnw_dec_job& j = jobs[i];
        if (j.n < 0 || j.bits < 0 || j.bits > 64) return fail(c, MNW_E_ARG, "decode: bits outside 0..64 or negative n");
        if (j.kind != MNW_INT_GROUP && j.kind != MNW_FLOAT_GROUP) return fail(c, MNW_E_ARG, "decode: unknown kind");
        if (j.n == 0) continue;
        if (!j.dst || (j.bits > 0 && !j.src)) return fail(c, MNW_E_ARG, "decode: null pointer");
        if (j.src_len < mnw::array_bytes(j.bits, j.n) && j.bits > 0) return fail(c, MNW_E_BUFLEN, "decode: stream shorter than ArrayBytes(bits, n)");
    }
    JobSlot* sl;
    int rc = acquire_slot(c, njobs > 0 ? njobs : 1, &sl, true);
    if (rc) return rc;
    // three launch classes: float 32-bit path, float 64-bit path, int64
    int cls_begin[4] = {0, 0, 0, 0};
    bool cls_xf[3] = {false, false, false};
    int nslot = 0;
    for (int cls = 0; cls < 3; ++cls) {
        cls_begin[cls] = nslot;
        int64_t tiles = 0;
        for (int i = 0; i < njobs; ++i) {
            const mnw_dec_job& j = jobs[i];
            if (j.n == 0) continue;
            int jc;
            if (j.kind == MNW_INT_GROUP) jc = 2;
            else {
                const bool narrow = j.bits <= 32 && j.pixels <= 0x7fffffffll && j.min >= -0x80000000ll &&
                                    j.min + (int64_t)((1ull << j.bits) - 1) <= 0x7fffffffll;
                jc = narrow ? 0 : 1;
            }
            if (jc != cls) continue;
            DevJob& d = sl->h_jobs[nslot++];
            memset(&d, 0, sizeof d);
            d.src = j.src; d.dst = j.dst; d.n = j.n; d.cap = j.src_len; d.pixels = j.pixels;
            d.min = j.min; d.bits = (int32_t)j.bits; d.seed = j.seed; d.block = j.block;
            d.low = j.low; d.high = j.high; d.kind = j.kind; d.periodic = j.periodic;
            d.mode = mnw::kModeExplicit;
            if (j.kind == MNW_FLOAT_GROUP) {
                d.dxq = quant_dx(j.low, j.high, j.pixels);   // group.go:305-306
                d.xform = c->unlog_fused ? (j.xform & MNW_XF_LOG10) : 0;   // minh's 10^x inside the unpack kernel
                cls_xf[cls] |= d.xform != 0;
            }
            d.tile0 = tiles;
            tiles += (j.n + mnw::kTileD - 1) / mnw::kTileD;
        }
    }
    cls_begin[3] = nslot;
    if (nslot == 0) return MNW_OK;
    const size_t table_bytes = sizeof(DevJob) * (size_t)nslot;
    if (!tables_unchanged(sl, 1, table_bytes)) {
        rc = prepare_slot(c, sl, s, nslot, false);
        if (rc) return rc;
        remember_tables(sl, 1, table_bytes);
    }
    for (int cls = 0; cls < 3; ++cls) {
        const int j0 = cls_begin[cls], j1 = cls_begin[cls + 1];
        if (j1 == j0) continue;
        int64_t tiles = sl->h_jobs[j1 - 1].tile0 + (sl->h_jobs[j1 - 1].n + mnw::kTileD - 1) / mnw::kTileD;
        const int grid = decode_grid_for(c, tiles);
        if (cls < 2) { ScopedTimer t(c, s, MNW_K_UNPACK_F32); mnw::launch_unpack_f32(s, grid, cls == 0, cls_xf[cls], sl->d_jobs + j0, j1 - j0, tiles); }
        else { ScopedTimer t(c, s, MNW_K_UNPACK_I64); mnw::launch_unpack_i64(s, grid, sl->d_jobs + j0, j1 - j0, tiles); }
    }
    HIPCHK(c, hipGetLastError());
    // minh's logarithmic columns: 10^x on the decoded values (go/minh/minh.go:315-319): inside the unpack kernel (above), or -- option
    // unlog_fused = 0 -- as a pass over the output
    if (!c->unlog_fused) {
        mnw::MinhArgs a;
        memset(&a, 0, sizeof a);
        int m = 0;
        int64_t max_n = 0;
        for (int i = 0; i < njobs; ++i) {
            const mnw_dec_job& j = jobs[i];
            if (!(j.xform & MNW_XF_LOG10) || j.kind != MNW_FLOAT_GROUP || j.n == 0) continue;
            mnw::MinhCol& col = a.col[m++];
            col.src = (const float*)j.dst; col.dst = (float*)j.dst; col.n = j.n; col.xform = mnw::kXfLog10;
            if (j.n > max_n) max_n = j.n;
            if (m == mnw::kMinhBatch) {
                { ScopedTimer t(c, s, MNW_K_MINH_UNLOG); mnw::launch_minh_unlog(s, c->cus, a, m, max_n); }
                m = 0; max_n = 0;
            }
        }
        if (m) { ScopedTimer t(c, s, MNW_K_MINH_UNLOG); mnw::launch_minh_unlog(s, c->cus, a, m, max_n); }
        HIPCHK(c, hipGetLastError());
    }
    return release_slot(c, sl, s);
}

// ---- minp vector blocks (go/minp/minp.go) ------------------------------------------------------
namespace {
int vectors_check(mnw_ctx* c, int64_t n_file, int64_t sub_cells, const char* who) {
    if (!c) return MNW_E_ARG;
    if (n_file < 1 || sub_cells < 1 || n_file % sub_cells != 0) return fail(c, MNW_E_ARG, who);
    if (n_file > 1290 || n_file * n_file * n_file >= (1ll << 31)) return fail(c, MNW_E_ARG, "vectors: n_file^3 must be below 2^31");
    return MNW_OK;
}
int ensure_soa(mnw_ctx* c, int64_t n) {
    void* p = c->d_soa;
    int rc = ensure_dev(c, &p, &c->cap_soa, (size_t)n * 3 * sizeof(float));
    c->d_soa = (float*)p;
    return rc;
}
}  // namespace

int mnw_subcell_gather(mnw_ctx* c, mnw_stream_t stream, const float* vec, int64_t n_file, int64_t sub_cells, float* soa) {
    int rc = vectors_check(c, n_file, sub_cells, "gather: n_file must be a positive multiple of sub_cells");
    if (rc) return rc;
    if (!vec || !soa) return fail(c, MNW_E_ARG, "gather: null pointer");
    HIPCHK(c, hipSetDevice(c->device));
    hipStream_t s = stream ? (hipStream_t)stream : c->stream;
    { ScopedTimer t(c, s, MNW_K_SUBCELL_GATHER); mnw::launch_subcell_gather(s, c->cus, vec, soa, n_file, sub_cells); }
    HIPCHK(c, hipGetLastError());
    return MNW_OK;
}

int mnw_subcell_scatter(mnw_ctx* c, mnw_stream_t stream, const float* soa, int64_t n_file, int64_t sub_cells, int periodic, float L,
                        float* vec) {
    int rc = vectors_check(c, n_file, sub_cells, "scatter: n_file must be a positive multiple of sub_cells");
    if (rc) return rc;
    if (!vec || !soa) return fail(c, MNW_E_ARG, "scatter: null pointer");
    HIPCHK(c, hipSetDevice(c->device));
    hipStream_t s = stream ? (hipStream_t)stream : c->stream;
    { ScopedTimer t(c, s, MNW_K_SUBCELL_SCATTER); mnw::launch_subcell_scatter(s, c->cus, soa, vec, n_file, sub_cells, periodic != 0, L); }
    HIPCHK(c, hipGetLastError());
    return MNW_OK;
}

int mnw_vectors_bounds(mnw_ctx* c, mnw_stream_t stream, const float* vec, int64_t n, float mn[3], float mx[3]) {
    if (!c || !vec || n < 1 || !mn || !mx) return fail(c, MNW_E_ARG, "bounds: bad arguments");   // bounds() reads vec[0]
    HIPCHK(c, hipSetDevice(c->device));
    hipStream_t s = stream ? (hipStream_t)stream : c->stream;
    if (!c->d_bkeys) {
        HIPCHK(c, hipMalloc((void**)&c->d_bkeys, 8 * sizeof(uint32_t)));
        HIPCHK(c, hipHostMalloc((void**)&c->h_bkeys, 8 * sizeof(uint32_t), hipHostMallocDefault));
    }
    HIPCHK(c, hipMemsetAsync(c->d_bkeys, 0, 8 * sizeof(uint32_t), s));
    { ScopedTimer t(c, s, MNW_K_BOUNDS3); mnw::launch_bounds3(s, c->cus, vec, n, c->d_bkeys); }
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, hipMemcpyAsync(c->h_bkeys, c->d_bkeys, 8 * sizeof(uint32_t), hipMemcpyDeviceToHost, s));
    HIPCHK(c, hipStreamSynchronize(s));
    auto unkey = [](uint32_t k) { uint32_t u = k ^ ((k >> 31) ? 0x80000000u : 0xffffffffu); float f; memcpy(&f, &u, 4); return f; };
    for (int k = 0; k < 3; ++k) {
        if (c->h_bkeys[2 * k] == 0 || c->h_bkeys[2 * k + 1] == 0) return fail(c, MNW_E_DOMAIN, "bounds: a component holds no ordered value (all NaN)");
        mx[k] = unkey(c->h_bkeys[2 * k]);
        mn[k] = unkey(~c->h_bkeys[2 * k + 1]);
    }
    return MNW_OK;
}

int mnw_vectors_encode(mnw_ctx* c, mnw_stream_t stream, const float* vec, int64_t n_file, int64_t sub_cells,
                       const float low[3], const float high[3], const int64_t pixels[3],
                       uint8_t* dst, int64_t slot_bytes, mnw_block_meta* meta) {
    int rc = vectors_check(c, n_file, sub_cells, "vectors: n_file must be a positive multiple of sub_cells");
    if (rc) return rc;
    if (!vec || !dst || !meta || !low || !high || !pixels || slot_bytes < 0 || (slot_bytes & 15) || ((uintptr_t)dst & 15))
        return fail(c, MNW_E_ARG, "vectors: null or misaligned pointer, or slot_bytes not a multiple of 16");
    const int64_t n = n_file * n_file * n_file, s3 = sub_cells * sub_cells * sub_cells, nsub3 = n / s3;
    HIPCHK(c, hipSetDevice(c->device));
    hipStream_t s = stream ? (hipStream_t)stream : c->stream;
    // ---- without staging: the reduce and the pack read the grid themselves (k_reduce_vec3 / k_pack_vec3) ----
    bool fused = c->vec3_fused && !c->pending && (n_file / sub_cells) % 4 == 0 && n_file % 4 == 0 && ((uintptr_t)vec & 15) == 0 &&
                 3 * s3 < (1 << 20);
    for (int k = 0; k < 3 && fused; ++k) {
        mnw_enc_job probe;
        memset(&probe, 0, sizeof probe);
        probe.low = low[k]; probe.high = high[k]; probe.pixels = pixels[k]; probe.kind = MNW_FLOAT_GROUP;
        fused = pixels[k] > 0 && pixels[k] <= 0x7fffffffll && float_fast_ok(probe);
    }
    if (fused) {
        JobSlot* sl;
        rc = acquire_slot(c, (int)(3 * s3), &sl);
        if (rc) return rc;
        for (int k = 0; k < 3; ++k) {
            for (int64_t sc = 0; sc < s3; ++sc) {
                mnw_enc_job j;
                memset(&j, 0, sizeof j);
                j.src = vec; j.dst = dst + (size_t)(k * s3 + sc) * slot_bytes; j.n = nsub3; j.cap = slot_bytes;
                j.kind = MNW_FLOAT_GROUP; j.low = low[k]; j.high = high[k]; j.pixels = pixels[k]; j.periodic = 1;
                fill_enc(sl->h_jobs[k * s3 + sc], j);
            }
        }
        rc = prepare_slot(c, sl, s, (int)(3 * s3), true);
        if (rc) return rc;
        const int64_t vt = mnw::vec3_tiles(n_file, sub_cells);
        // ---- the grid read once (k_encode_reg_vec3): every block at most G vtiles for a resident grid of G workgroups ----
        bool two_pass = true;
        bool v3_backing_off = false;
        if (c->vec3_single_read && c->single_read_backoff && c->sr_backoff_left > 0) {
            --c->sr_backoff_left;
            c->counters[MNW_C_BACKOFF_BATCHES]++;
            v3_backing_off = true;
        }
        if (c->vec3_single_read && !v3_backing_off) {
            if (c->reg_vec3_blocks_per_cu == 0) {
                const int occ = mnw::encode_reg_vec3_blocks_per_cu(c->vec3_deep);
                c->reg_vec3_blocks_per_cu = occ < 1 ? -1 : (occ > 3 ? 3 : occ);
            }
            int64_t grid = (int64_t)c->cus * c->reg_vec3_blocks_per_cu;
            if (c->grid_cap > 0 && c->grid_cap < grid) grid = c->grid_cap;
            // (the last vec3_reg_sweepers() workgroups of the grid hold no items: they derive the sub-cells' descriptors)
            const int64_t nsw = mnw::vec3_reg_sweepers();
            if (vt + nsw < grid) grid = vt + nsw;
            else grid -= (grid - nsw) % 3;               // (the item workgroups in three equal thirds: the weighted hand-out)
            const int64_t T = vt / s3;
            if (c->reg_vec3_blocks_per_cu >= 1 && T <= grid - nsw && T <= mnw::vec3_reg_max_tiles_per_block() && mnw::vec3_reg_geometry_ok(n_file, sub_cells)) {
                const size_t need = (size_t)mnw::vec3_reg_recs(n_file, sub_cells) * mnw::item_rec_bytes();
                if (need > c->cap_recs) {
                    if (c->d_recs) { HIPCHK(c, hipStreamSynchronize(s)); HIPCHK(c, hipFree(c->d_recs)); }
                    c->d_recs = nullptr; c->cap_recs = 0;
                    const size_t want = need + need / 2 + 4096;
                    HIPCHK(c, hipMalloc(&c->d_recs, want));
                    HIPCHK(c, hipMemsetAsync(c->d_recs, 0, want, s));      // tags of a fresh buffer must not match any generation
                    c->cap_recs = want;
                }
                c->reg_gen = (c->reg_gen + 1) & 0x7fffffffu;
                if (c->reg_gen == 0) c->reg_gen = 1;
                sl->h_sync[0] = 0;
                { ScopedTimer t(c, s, MNW_K_ENCODE_REG_VEC3);
                  mnw::launch_encode_reg_vec3(s, (int)grid, vec, n_file, sub_cells, sl->d_jobs, c->d_recs, c->reg_gen, abort_ptr(sl), sl->d_hsync,
                                              sl->d_meta, c->reg_drop_item, c->vec3_round_table, c->vec3_deep); }
                HIPCHK(c, hipGetLastError());
                HIPCHK(c, wait_stream(s));
                c->counters[MNW_C_VEC3_SINGLE_READ_CALLS]++;
                two_pass = false;
                if (sl->h_sync[0] == 0) c->sr_backoff = 0;
                if (sl->h_sync[0] != 0) {
                    // gave up (something else held part of the GPU): the two launches below redo the call
                    c->counters[MNW_C_SINGLE_READ_ABORTS]++;
                    c->sr_backoff = c->sr_backoff ? (c->sr_backoff < 64 ? 2 * c->sr_backoff : 64) : 1;
                    c->sr_backoff_left = c->sr_backoff;
                    rc = prepare_slot(c, sl, s, (int)(3 * s3), true);
                    if (rc) return rc;
                    two_pass = true;
                }
            }
        }
        if (two_pass) {
        int64_t grl = c->vec3_tiles_per_wg > 0 ? (vt + c->vec3_tiles_per_wg - 1) / c->vec3_tiles_per_wg : (int64_t)c->cus * 8;
        int64_t gpl = c->vec3_tiles_per_wg > 0 ? (vt + c->vec3_tiles_per_wg - 1) / c->vec3_tiles_per_wg : (int64_t)c->cus * 4;
        int gr = (int)(grl > vt ? vt : grl), gp = (int)(gpl > vt ? vt : gpl);
        { ScopedTimer t(c, s, MNW_K_REDUCE_VEC3); mnw::launch_reduce_vec3(s, gr, vec, n_file, sub_cells, sl->d_jobs, sl->d_stats); }
        { ScopedTimer t(c, s, MNW_K_PACK_VEC3); mnw::launch_pack_vec3(s, gp, vec, n_file, sub_cells, sl->d_jobs, sl->d_stats, sl->d_meta); }
        HIPCHK(c, hipGetLastError());
        HIPCHK(c, wait_stream(s));
        }
        bool exact = false;
        int first = MNW_OK;
        for (int64_t b = 0; b < 3 * s3; ++b) {
            const Meta& m = sl->h_meta[b];
            if (m.status == mnw::kNeedsExact) exact = true;
            else if (m.status != 0 && first == MNW_OK) first = (int)m.status;
        }
        if (!exact) {
            memcpy(meta, sl->h_meta, sizeof(Meta) * (size_t)(3 * s3));
            if (first != MNW_OK) fail(c, first, mnw_strerror(first));
            return first;
        }
        // some block is order dependent (NaN, values outside [low, high), ...): redo the call through the staging
        // path, whose block encoder has the exact path behind it
    }
    rc = ensure_soa(c, n);
    if (rc) return rc;
    { ScopedTimer t(c, s, MNW_K_SUBCELL_GATHER); mnw::launch_subcell_gather(s, c->cus, vec, c->d_soa, n_file, sub_cells); }
    HIPCHK(c, hipGetLastError());
    std::vector<mnw_enc_job> jobs((size_t)(3 * s3));
    for (int k = 0; k < 3; ++k) {
        for (int64_t sc = 0; sc < s3; ++sc) {
            mnw_enc_job& j = jobs[(size_t)(k * s3 + sc)];
            memset(&j, 0, sizeof j);
            j.src = c->d_soa + (size_t)k * n + (size_t)sc * nsub3;
            j.dst = dst + (size_t)(k * s3 + sc) * slot_bytes;
            j.n = nsub3; j.cap = slot_bytes;
            j.kind = MNW_FLOAT_GROUP; j.low = low[k]; j.high = high[k]; j.pixels = pixels[k];
            j.periodic = 1;                                   // Writer.FloatGroup, writer.go:74
        }
    }
    return mnw_encode_blocks(c, stream, jobs.data(), (int)jobs.size(), meta);
}

int mnw_vectors_decode(mnw_ctx* c, mnw_stream_t stream, const uint8_t* src, int64_t slot_bytes, const mnw_block_meta* meta,
                       int64_t n_file, int64_t sub_cells, const float low[3], const float high[3], const int64_t pixels[3],
                       int periodic, float L, uint64_t seed, uint64_t first_block, float* vec_out) {
    int rc = vectors_check(c, n_file, sub_cells, "vectors: n_file must be a positive multiple of sub_cells");
    if (rc) return rc;
    if (!src || !vec_out || !meta || !low || !high || !pixels || slot_bytes < 0)
        return fail(c, MNW_E_ARG, "vectors: null pointer");
    const int64_t n = n_file * n_file * n_file, s3 = sub_cells * sub_cells * sub_cells, nsub3 = n / s3;
    HIPCHK(c, hipSetDevice(c->device));
    hipStream_t s = stream ? (hipStream_t)stream : c->stream;
    // ---- without staging: k_unpack_vec3 writes the grid itself --------------------------------------------
    bool fused = c->vec3_fused && !c->pending && (n_file / sub_cells) % 4 == 0 && n_file % 4 == 0 && ((uintptr_t)vec_out & 15) == 0 &&
                 3 * s3 < (1 << 20) && mnw::vec3_unpack_geometry_ok(n_file, sub_cells);
    int max_bits = 0;
    for (int64_t b = 0; b < 3 * s3 && fused; ++b) {
        const int k = (int)(b / s3);
        const mnw_block_meta& m = meta[b];
        if (m.bits > max_bits) max_bits = (int)m.bits;
        if (m.bits < 0 || m.bits > 64) return fail(c, MNW_E_ARG, "decode: bits outside 0..64");
        if (m.bits > 0 && m.nbytes < mnw::array_bytes(m.bits, nsub3)) return fail(c, MNW_E_BUFLEN, "decode: stream shorter than ArrayBytes(bits, n)");
        // (a negative minimum -- a block the exact path made of out-of-range data -- needs bound() as written: staging path)
        fused = m.bits <= 32 && pixels[k] > 0 && pixels[k] <= 0x7fffffffll && m.min >= 0 &&
                m.min + (int64_t)((1ull << m.bits) - 1) <= 0x7fffffffll;
    }
    if (fused) {
        JobSlot* sl;
        rc = acquire_slot(c, (int)(3 * s3), &sl, true);
        if (rc) return rc;
        for (int k = 0; k < 3; ++k) {
            const float dxq = quant_dx(low[k], high[k], pixels[k]);                 // group.go:305-306
            for (int64_t sc = 0; sc < s3; ++sc) {
                const size_t b = (size_t)(k * s3 + sc);
                DevJob& d = sl->h_jobs[b];
                memset(&d, 0, sizeof d);
                d.src = src + b * (size_t)slot_bytes; d.dst = vec_out; d.n = nsub3; d.cap = meta[b].nbytes; d.pixels = pixels[k];
                d.min = meta[b].min; d.bits = (int32_t)meta[b].bits; d.seed = seed; d.block = first_block + b;
                d.low = low[k]; d.high = high[k]; d.kind = MNW_FLOAT_GROUP; d.periodic = 1; d.mode = mnw::kModeExplicit;
                d.dxq = dxq;
            }
        }
        const size_t table_bytes = sizeof(DevJob) * (size_t)(3 * s3);
        if (!tables_unchanged(sl, 1, table_bytes)) {
            rc = prepare_slot(c, sl, s, (int)(3 * s3), false);
            if (rc) return rc;
            remember_tables(sl, 1, table_bytes);
        }
        const int64_t vt = mnw::vec3_tiles(n_file, sub_cells);
        int64_t gl = c->vec3_tiles_per_wg > 0 ? (vt + c->vec3_tiles_per_wg - 1) / c->vec3_tiles_per_wg : (int64_t)c->cus * 8;
        int g = (int)(gl > vt ? vt : gl);
        { ScopedTimer t(c, s, MNW_K_UNPACK_VEC3); mnw::launch_unpack_vec3(s, g, vec_out, n_file, sub_cells, sl->d_jobs, periodic != 0, L, max_bits, c->vec3_unpack_db != 0,
                                                                                   c->vec3_x_order ? c->vec3_unpack_tiles : 0, c->vec3_x_order >= 2); }
        HIPCHK(c, hipGetLastError());
        return release_slot(c, sl, s);
    }
    rc = ensure_soa(c, n);
    if (rc) return rc;
    std::vector<mnw_dec_job> jobs((size_t)(3 * s3));
    for (int k = 0; k < 3; ++k) {
        for (int64_t sc = 0; sc < s3; ++sc) {
            const size_t b = (size_t)(k * s3 + sc);
            mnw_dec_job& j = jobs[b];
            memset(&j, 0, sizeof j);
            j.src = src + b * (size_t)slot_bytes;
            j.dst = c->d_soa + (size_t)k * n + (size_t)sc * nsub3;
            j.n = nsub3; j.src_len = meta[b].nbytes; j.min = meta[b].min; j.bits = meta[b].bits;
            j.kind = MNW_FLOAT_GROUP; j.low = low[k]; j.high = high[k]; j.pixels = pixels[k]; j.periodic = 1;
            j.seed = seed; j.block = first_block + b;
        }
    }
    rc = mnw_decode_blocks(c, stream, jobs.data(), (int)jobs.size());
    if (rc) return rc;
    { ScopedTimer t(c, s, MNW_K_SUBCELL_SCATTER); mnw::launch_subcell_scatter(s, c->cus, c->d_soa, vec_out, n_file, sub_cells, periodic != 0, L); }
    HIPCHK(c, hipGetLastError());
    return MNW_OK;
}

// ---- minh column transforms (go/minh/minh.go:139-149, 315-319) ------------------------------------
namespace {
int minh_launch(mnw_ctx* c, mnw_stream_t stream, const mnw_minh_col* cols, int ncols, bool unlog) {
    if (!c || ncols < 0 || (ncols > 0 && !cols)) return fail(c, MNW_E_ARG, "minh: bad arguments");
    HIPCHK(c, hipSetDevice(c->device));
    hipStream_t s = stream ? (hipStream_t)stream : c->stream;
    for (int c0 = 0; c0 < ncols; c0 += mnw::kMinhBatch) {
        mnw::MinhArgs a;
        memset(&a, 0, sizeof a);
        int m = 0;
        int64_t max_n = 0;
        for (int i = c0; i < ncols && m < mnw::kMinhBatch; ++i) {
            const mnw_minh_col& col = cols[i];
            if (col.n < 0 || (col.n > 0 && (!col.dst || (!unlog && !col.src)))) return fail(c, MNW_E_ARG, "minh: null pointer or negative length");
            if (col.n == 0) continue;
            mnw::MinhCol& d = a.col[m++];
            d.src = unlog ? col.dst : col.src; d.dst = col.dst; d.n = col.n; d.low = col.low; d.high = col.high;
            d.high_below = nextafterf(col.high, -INFINITY);       // math.Nextafter32(col.High, float32(math.Inf(-1)))
            d.xform = unlog ? mnw::kXfLog10 : ((col.log ? mnw::kXfLog10 : 0) | mnw::kXfClamp);
            if (col.n > max_n) max_n = col.n;
        }
        if (m == 0) continue;
        if (unlog) { ScopedTimer t(c, s, MNW_K_MINH_UNLOG); mnw::launch_minh_unlog(s, c->cus, a, m, max_n); }
        else { ScopedTimer t(c, s, MNW_K_MINH_PROCESS); mnw::launch_minh_process(s, c->cus, a, m, max_n); }
        HIPCHK(c, hipGetLastError());
    }
    return MNW_OK;
}
}  // namespace

int mnw_minh_selftest(mnw_ctx* c, int which, uint64_t first, uint64_t count, uint64_t out[2]) {
    if (!c || !out || which < 0 || which > 1) return fail(c, MNW_E_ARG, "selftest: bad arguments");
    HIPCHK(c, hipSetDevice(c->device));
    unsigned long long* d = nullptr;
    HIPCHK(c, hipMalloc((void**)&d, 16));
    HIPCHK(c, hipMemsetAsync(d, 0, 16, c->stream));
    mnw::launch_minh_selftest(c->stream, which, first, count, d);
    unsigned long long h[2] = {0, 0};
    hipError_t e = hipMemcpyAsync(h, d, 16, hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    hipFree(d);
    if (e != hipSuccess) return fail(c, MNW_E_HIP, "selftest", e);
    out[0] = h[0]; out[1] = h[1];
    return MNW_OK;
}

int mnw_minh_process(mnw_ctx* c, mnw_stream_t stream, const mnw_minh_col* cols, int ncols) { return minh_launch(c, stream, cols, ncols, false); }
int mnw_minh_unlog(mnw_ctx* c, mnw_stream_t stream, const mnw_minh_col* cols, int ncols) { return minh_launch(c, stream, cols, ncols, true); }

// ---- text catalogue ingest (go/text/text.go:176-197, go/text/parse.go) --------------------------------
namespace {
constexpr unsigned kTextSlowCap0 = 1u << 16;      // first size of the list of literals left to the host; it grows on demand

bool go_is_space(uint8_t ch) {      // unicode.IsSpace for the Latin-1 range, as bytes.Fields sees single bytes
    return ch == ' ' || (ch >= '\t' && ch <= '\r') || ch == 0x85 || ch == 0xA0;
}

// strconv.ParseFloat(tok, 64) for the literals the device leaves alone (Go 1.16, strconv/atof.go: special, readFloat,
// underscoreOK restated; the digits themselves go through strtod in the "C" locale, correctly rounded like ParseFloat).
// 0: *out is the value, 2: syntax error, 3: value out of range (ErrRange: the reference panics on both).
int go_parse_float(const std::string& tok, double* out) {
    const char* s = tok.c_str();
    const size_t n = tok.size();
    auto lower = [](char ch) { return (char)(ch | 0x20); };
    if (n == 0) return 2;
    // special(): an optional sign only in front of inf / infinity, none in front of nan; the whole token must be consumed
    {
        size_t i = 0;
        int sign = 1;
        bool had_sign = false;
        if (s[0] == '+' || s[0] == '-') { sign = s[0] == '-' ? -1 : 1; had_sign = true; i = 1; }
        auto prefix = [&](const char* w) { size_t k = 0; while (w[k] && i + k < n && lower(s[i + k]) == w[k]) ++k; return k; };
        if (i < n && lower(s[i]) == 'i') {
            size_t k = prefix("infinity");
            if (k > 3 && k < 8) k = 3;
            if ((k == 3 || k == 8) && i + k == n) { *out = sign * INFINITY; return 0; }
            return 2;
        }
        if (i < n && lower(s[i]) == 'n') {
            if (!had_sign && prefix("nan") == 3 && n == 3) { *out = NAN; return 0; }
            return 2;
        }
    }
    // readFloat's grammar: sign, optional 0x, digits with at most one '.', optional exponent (e for decimal; p, mandatory, for
    // hexadecimal) with an optional sign and at least one digit; '_' anywhere in between, judged by underscoreOK
    size_t i = 0;
    if (s[i] == '+' || s[i] == '-') ++i;
    bool hex = false;
    if (i + 2 < n && s[i] == '0' && lower(s[i + 1]) == 'x') { hex = true; i += 2; }
    bool sawdot = false, sawdigits = false, sawexp = false, underscores = false;
    for (; i < n; ++i) {
        const char ch = s[i];
        if (ch == '_') { underscores = true; continue; }
        if (ch == '.') { if (sawdot) break; sawdot = true; continue; }
        if ((ch >= '0' && ch <= '9') || (hex && lower(ch) >= 'a' && lower(ch) <= 'f')) { sawdigits = true; continue; }
        break;
    }
    if (!sawdigits) return 2;
    if (i < n && lower(s[i]) == (hex ? 'p' : 'e')) {
        ++i;
        if (i >= n) return 2;
        if (s[i] == '+' || s[i] == '-') ++i;
        if (i >= n || s[i] < '0' || s[i] > '9') return 2;
        for (; i < n && ((s[i] >= '0' && s[i] <= '9') || s[i] == '_'); ++i) if (s[i] == '_') underscores = true;
        sawexp = true;
    }
    if (i != n) return 2;
    if (hex && !sawexp) return 2;
    std::string digits = tok;
    if (underscores) {
        // underscoreOK: an underscore only between digits, or between the base prefix and a digit
        char saw = '^';
        size_t k = 0;
        if (s[k] == '+' || s[k] == '-') ++k;
        bool hx = false;
        if (n - k >= 2 && s[k] == '0' && (lower(s[k + 1]) == 'b' || lower(s[k + 1]) == 'o' || lower(s[k + 1]) == 'x')) { hx = lower(s[k + 1]) == 'x'; k += 2; saw = '0'; }
        for (; k < n; ++k) {
            const char ch = s[k];
            if ((ch >= '0' && ch <= '9') || (hx && lower(ch) >= 'a' && lower(ch) <= 'f')) { saw = '0'; continue; }
            if (ch == '_') { if (saw != '0') return 2; saw = '_'; continue; }
            if (saw == '_') return 2;
            saw = '!';
        }
        if (saw == '_') return 2;
        digits.clear();
        for (char ch : tok) if (ch != '_') digits.push_back(ch);
    }
    static locale_t c_locale = newlocale(LC_ALL_MASK, "C", (locale_t)0);      // a host program may have set a comma-decimal locale
    char* endp = nullptr;
    const double v = c_locale ? strtod_l(digits.c_str(), &endp, c_locale) : strtod(digits.c_str(), &endp);
    if (endp != digits.c_str() + digits.size()) return 2;
    if (isinf(v)) return 3;
    *out = v;
    return 0;
}
}  // namespace

namespace {
// the text on the device, lines / kept lines per tile counted and prefixed, totals on the host
int text_load(mnw_ctx* c, const uint8_t* text, int64_t nbytes, uint8_t sep, uint8_t comment, bool force, const void* resident = nullptr) {
    if (!force && c->text_res == text && c->text_res_n == nbytes && c->text_res_sep == sep && c->text_res_comm == comment) return MNW_OK;
    c->text_res = nullptr;
    HIPCHK(c, hipSetDevice(c->device));
    hipStream_t s = c->stream;
    const int64_t tiles = (nbytes + mnw::kTextTile - 1) / mnw::kTextTile;
    int rc = MNW_OK;
    if (!resident) {
        rc = ensure_dev(c, &c->d_text, &c->cap_text, (size_t)nbytes + 64);
        if (rc) return rc;
    }
    rc = ensure_dev(c, &c->d_tile, &c->cap_tile, (size_t)tiles * 16 + 32 + (size_t)((tiles + 1023) / 1024 + 1) * 16);     // counters, then the scan's chunk totals
    if (rc) return rc;
    if (!c->d_tslow) { HIPCHK(c, hipMalloc(&c->d_tslow, sizeof(mnw::TextSlow) * kTextSlowCap0)); c->cap_tslow = kTextSlowCap0; }
    if (!c->d_tstate) HIPCHK(c, hipMalloc(&c->d_tstate, sizeof(mnw::TextState)));
    if (resident) c->d_text_now = resident;                    // the caller has uploaded the block (another context, another stream)
    else {
        rc = copy_h2d(c, s, c->d_text, text, (size_t)nbytes);  // pinned, chunked, the host-side copy on several threads
        if (rc) return rc;
        c->d_text_now = c->d_text;
    }
    {
        ScopedTimer t(c, s, MNW_K_TEXT_LINES);
        mnw::launch_text_count(s, (const uint8_t*)c->d_text_now, nbytes, sep, comment, (int64_t*)c->d_tile);
        mnw::launch_text_scan(s, (int64_t*)c->d_tile, tiles, (mnw::TextState*)c->d_tstate);
    }
    HIPCHK(c, hipGetLastError());
    mnw::TextState st;
    HIPCHK(c, hipMemcpyAsync(&st, c->d_tstate, sizeof st, hipMemcpyDeviceToHost, s));
    HIPCHK(c, hipStreamSynchronize(s));
    c->text_res = text; c->text_res_n = nbytes; c->text_res_sep = sep; c->text_res_comm = comment;
    c->text_res_lines = st.n_lines; c->text_res_rows = st.n_rows;
    return MNW_OK;
}
}  // namespace

int mnw_text_rows(mnw_ctx* c, const uint8_t* text, int64_t nbytes, uint8_t sep, uint8_t comment, int64_t* n_lines, int64_t* n_rows) {
    if (!c || nbytes < 0 || (nbytes > 0 && !text)) return fail(c, MNW_E_ARG, "text: bad arguments");
    if (sep == '\n' || comment == '\n') return fail(c, MNW_E_ARG, "text: the newline cannot be the separator or the comment character");
    if (n_lines) *n_lines = nbytes == 0 ? 1 : 0;
    if (n_rows) *n_rows = 0;
    if (nbytes == 0) return MNW_OK;
    const int rc = text_load(c, text, nbytes, sep, comment, true);
    if (rc) return rc;
    if (n_lines) *n_lines = c->text_res_lines;
    if (n_rows) *n_rows = c->text_res_rows;
    return MNW_OK;
}

namespace {
// One pass of k_text_parse over the block that is on the device (text_load) for one set of columns in which every field
// appears at most once, then the literals the device left to the host.
int text_parse_pass(mnw_ctx* c, hipStream_t s, const uint8_t* text, int64_t nbytes, uint8_t sep, uint8_t comment, mnw::TextCols& cols,
                    int64_t capacity_rows, int64_t* n_rows) {
    mnw::TextState st;
    for (int attempt = 0; attempt < 2; ++attempt) {
        mnw::TextState st0;
        memset(&st0, 0, sizeof st0);
        st0.err_line = ~0ull;
        st0.n_rows = c->text_res_rows; st0.n_lines = c->text_res_lines;
        HIPCHK(c, hipMemcpyAsync(c->d_tstate, &st0, sizeof st0, hipMemcpyHostToDevice, s));
        {
            ScopedTimer t(c, s, MNW_K_TEXT_PARSE);
            mnw::launch_text_parse(s, (const uint8_t*)c->d_text_now, nbytes, sep, comment, (const int64_t*)c->d_tile, cols, capacity_rows,
                                   (mnw::TextSlow*)c->d_tslow, (unsigned)c->cap_tslow, (mnw::TextState*)c->d_tstate);
        }
        HIPCHK(c, hipGetLastError());
        HIPCHK(c, hipMemcpyAsync(&st, c->d_tstate, sizeof st, hipMemcpyDeviceToHost, s));
        HIPCHK(c, hipStreamSynchronize(s));
        if (st.n_slow <= c->cap_tslow) break;
        // more literals for the host than the list holds (the kernel counted them all): a list of that size, and the pass again
        if (attempt == 1) return fail(c, MNW_E_HIP, "text: the list of literals for the host overflowed twice");
        HIPCHK(c, hipFree(c->d_tslow));
        c->d_tslow = nullptr; c->cap_tslow = 0;
        const size_t want = (size_t)st.n_slow + (size_t)st.n_slow / 8 + 1024;
        if (hipMalloc(&c->d_tslow, sizeof(mnw::TextSlow) * want) != hipSuccess) { (void)hipGetLastError(); return fail(c, MNW_E_NOMEM, "text: no memory for the list of literals outside the device's fast path"); }
        c->cap_tslow = want;
    }
    c->text_stats[0] = st.n_lines; c->text_stats[1] = st.n_rows; c->text_stats[2] = st.tiles_general;
    *n_rows = st.n_rows;
    if (st.n_rows > capacity_rows || (st.flags & 1u)) return fail(c, MNW_E_CAPACITY, "text: more lines than capacity_rows");
    if (st.err_line != ~0ull) {
        char buf[200];
        const long long line = (long long)(st.err_line >> 8);
        const int code = (int)(st.err_line & 0xff);
        if (code == 1) snprintf(buf, sizeof buf, "Data on line %lld does not have %d columns.", line, cols.fields_expected);   // parse.go:114-118
        else snprintf(buf, sizeof buf, "line %lld: strconv.%s: invalid syntax or value out of range", line, code == 2 ? "Atoi" : "ParseFloat");
        return fail(c, MNW_E_DOMAIN, buf);
    }
    // ---- literals the device left alone (more than 19 digits, subnormal or overflowing values, inf / nan, hex floats,
    // digit separators): converted here with Go's rules, written back in one copy and one small launch
    if (st.n_slow > 0) {
        std::vector<mnw::TextSlow> sl(st.n_slow);
        HIPCHK(c, hipMemcpy(sl.data(), c->d_tslow, sizeof(mnw::TextSlow) * st.n_slow, hipMemcpyDeviceToHost));
        for (mnw::TextSlow& w : sl) {
            const std::string tok((const char*)text + w.offset, (size_t)w.len);
            double v = 0;
            const int rc = go_parse_float(tok, &v);
            if (rc) {
                char buf[200];
                snprintf(buf, sizeof buf, "strconv.ParseFloat: parsing \"%.60s\": %s", tok.c_str(), rc == 3 ? "value out of range" : "invalid syntax");
                return fail(c, MNW_E_DOMAIN, buf);
            }
            const float f = (float)v;                         // float32(x): beyond float32's range becomes +-Inf as in Go
            uint32_t u;
            memcpy(&u, &f, 4);
            w.offset = (int64_t)u;
        }
        HIPCHK(c, hipMemcpyAsync(c->d_tslow, sl.data(), sizeof(mnw::TextSlow) * st.n_slow, hipMemcpyHostToDevice, s));
        mnw::launch_text_patch(s, (const mnw::TextSlow*)c->d_tslow, st.n_slow, cols);
        HIPCHK(c, hipGetLastError());
        HIPCHK(c, hipStreamSynchronize(s));                   // (sl is pageable host memory the copy reads)
    }
    return MNW_OK;
}
}  // namespace

int mnw_text_rows_resident(mnw_ctx* c, const uint8_t* text, const uint8_t* text_dev, int64_t nbytes, uint8_t sep, uint8_t comment,
                           int64_t* n_lines, int64_t* n_rows) {
    if (!c || nbytes < 0 || (nbytes > 0 && (!text || !text_dev))) return fail(c, MNW_E_ARG, "text: bad arguments");
    if (sep == '\n' || comment == '\n') return fail(c, MNW_E_ARG, "text: the newline cannot be the separator or the comment character");
    if (n_lines) *n_lines = nbytes == 0 ? 1 : 0;
    if (n_rows) *n_rows = 0;
    if (nbytes == 0) return MNW_OK;
    const int rc = text_load(c, text, nbytes, sep, comment, true, text_dev);
    if (rc) return rc;
    if (n_lines) *n_lines = c->text_res_lines;
    if (n_rows) *n_rows = c->text_res_rows;
    return MNW_OK;
}

int mnw_text_parse(mnw_ctx* c, const uint8_t* text, int64_t nbytes, uint8_t sep, uint8_t comment,
                   const int32_t* float_cols, int n_float, float* const* float_out,
                   const int32_t* int_cols, int n_int, int64_t* const* int_out, int64_t capacity_rows, int64_t* n_rows) {
    if (!c || !n_rows || nbytes < 0 || n_float < 0 || n_int < 0 || (nbytes > 0 && !text)) return fail(c, MNW_E_ARG, "text: bad arguments");
    if (n_float + n_int > mnw::kTextMaxOut) return fail(c, MNW_E_ARG, "text: at most 128 columns per call");
    if ((n_float > 0 && (!float_cols || !float_out)) || (n_int > 0 && (!int_cols || !int_out))) return fail(c, MNW_E_ARG, "text: null column arrays");
    if (sep == '\n' || comment == '\n') return fail(c, MNW_E_ARG, "text: the newline cannot be the separator or the comment character");
    *n_rows = 0;
    if (nbytes == 0) return MNW_OK;
    int max_field = -1;
    for (int k = 0; k < n_float + n_int; ++k) {
        const bool isf = k < n_float;
        const int32_t f = isf ? float_cols[k] : int_cols[k - n_float];
        if (f < 0 || f >= mnw::kTextMaxCols) return fail(c, MNW_E_ARG, "text: field numbers must be in 0..255");
        if (!(isf ? (void*)float_out[k] : (void*)int_out[k - n_float])) return fail(c, MNW_E_ARG, "text: null output");
        if (f > max_field) max_field = f;
    }
    // bufLen = len(bytes.Fields(lines[0])) (parse.go:96): words of the first kept line, split at white space
    int words = -1;
    {
        int64_t p = 0;
        while (p <= nbytes && words < 0) {
            int64_t e = p;
            while (e < nbytes && text[e] != '\n') ++e;
            int64_t ce = p;
            while (ce < e && text[ce] != comment) ++ce;
            bool keep = false;
            for (int64_t q = p; q < ce; ++q) if (text[q] != sep) { keep = true; break; }
            if (keep) {
                words = 0;
                bool in = false;
                for (int64_t q = p; q < ce; ++q) { const bool sp = go_is_space(text[q]); if (!sp && !in) ++words; in = !sp; }
            }
            p = e + 1;
        }
        if (words < 0) {                                            // nothing but comments and blank lines
            int64_t nl = 1;
            for (int64_t q = 0; q < nbytes; ++q) nl += text[q] == '\n';
            c->text_stats[0] = nl; c->text_stats[1] = 0; c->text_stats[2] = 0;
            c->text_res = nullptr;
            return MNW_OK;
        }
        if (max_field >= words) {
            char buf[160];
            snprintf(buf, sizeof buf, "Data has %d columns, but column %d was requested.", words, max_field);    // parse.go:98-103
            return fail(c, MNW_E_DOMAIN, buf);
        }
    }
    HIPCHK(c, hipSetDevice(c->device));
    hipStream_t s = c->stream;
    int rc = text_load(c, text, nbytes, sep, comment, false);       // a no-op right after mnw_text_rows on the same buffer
    if (rc) return rc;
    c->text_res = nullptr;                                          // one use only: the caller may refill the same buffer with the next block
    // The reference's Block fills every requested output, also when a column is asked for twice or both as []int64 and as
    // []float32 (text.go:202-221 splits the request by type and parses each list on its own).  The kernel maps a field to one
    // output, so the outputs are dealt into passes in each of which a field appears once: one pass for any ordinary request.
    const int n_out = n_float + n_int;
    std::vector<char> done((size_t)(n_out > 0 ? n_out : 1), 0);
    int left = n_out;
    bool first = true;
    while (first || left > 0) {
        first = false;
        mnw::TextCols cols;
        memset(&cols, 0, sizeof cols);
        for (int k = 0; k < mnw::kTextMaxCols; ++k) cols.slot[k] = -1;
        cols.max_field = -1;
        // (the kernel numbers the float outputs first, then the int outputs)
        std::vector<int> pick_f, pick_i;
        for (int k = 0; k < n_out; ++k) {
            if (done[k]) continue;
            const bool isf = k < n_float;
            const int32_t f = isf ? float_cols[k] : int_cols[k - n_float];
            if (cols.slot[f] != -1) continue;                        // this field already has an output in this pass: next pass
            cols.slot[f] = 0;                                         // (marks the field taken; the output number is set below)
            (isf ? pick_f : pick_i).push_back(k);
            done[k] = 1; --left;
        }
        cols.n_float = (int)pick_f.size(); cols.n_int = (int)pick_i.size();
        int o = 0;
        for (int k : pick_f) { cols.slot[float_cols[k]] = (int16_t)o; cols.out[o] = (void*)float_out[k]; if (float_cols[k] > cols.max_field) cols.max_field = float_cols[k]; ++o; }
        for (int k : pick_i) { cols.slot[int_cols[k - n_float]] = (int16_t)o; cols.out[o] = (void*)int_out[k - n_float]; if (int_cols[k - n_float] > cols.max_field) cols.max_field = int_cols[k - n_float]; ++o; }
        cols.fields_expected = n_out > 0 ? words : -1;               // parseInt64s / parseFloat32s return at once when they have no column (parse.go:83, 132)
        rc = text_parse_pass(c, s, text, nbytes, sep, comment, cols, capacity_rows, n_rows);
        if (rc) return rc;
    }
    return MNW_OK;
}

// ---- device memory for callers without a HIP binding ---------------------------------------------------
int mnw_dev_alloc(mnw_ctx* c, int64_t nbytes, void** p) {
    if (!c || !p || nbytes < 0) return fail(c, MNW_E_ARG, "dev_alloc: bad arguments");
    *p = nullptr;
    HIPCHK(c, hipSetDevice(c->device));
    if (hipMalloc(p, (size_t)(nbytes > 0 ? nbytes : 1)) != hipSuccess) { (void)hipGetLastError(); return fail(c, MNW_E_NOMEM, "dev_alloc: out of device memory"); }
    return MNW_OK;
}
int mnw_dev_free(mnw_ctx* c, void* p) {
    if (!c) return MNW_E_ARG;
    if (p) { HIPCHK(c, hipSetDevice(c->device)); HIPCHK(c, hipFree(p)); }
    return MNW_OK;
}
int mnw_copy_to_device(mnw_ctx* c, void* dst, const void* src, int64_t nbytes) {
    if (!c || nbytes < 0 || (nbytes > 0 && (!dst || !src))) return fail(c, MNW_E_ARG, "copy_to_device: bad arguments");
    if (nbytes == 0) return MNW_OK;
    HIPCHK(c, hipSetDevice(c->device));
    int rc = copy_h2d(c, c->stream, dst, src, (size_t)nbytes);
    if (rc) return rc;
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return MNW_OK;
}
int mnw_copy_file_to_device(mnw_ctx* c, void* dst, int fd, int n, const int64_t* file_offsets, const int64_t* nbytes, const int64_t* dst_offsets) {
    if (!c || n < 0 || fd < 0 || (n > 0 && (!dst || !file_offsets || !nbytes))) return fail(c, MNW_E_ARG, "copy_file_to_device: bad arguments");
    for (int i = 0; i < n; ++i)
        if (nbytes[i] < 0 || file_offsets[i] < 0 || (dst_offsets && dst_offsets[i] < 0)) return fail(c, MNW_E_ARG, "copy_file_to_device: negative offset or size");
    HIPCHK(c, hipSetDevice(c->device));
    int rc = copy_file_h2d(c, c->stream, dst, fd, n, file_offsets, nbytes, dst_offsets);
    if (rc) return rc;
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return MNW_OK;
}
int mnw_copy_to_host(mnw_ctx* c, void* dst, const void* src, int64_t nbytes) {
    if (!c || nbytes < 0 || (nbytes > 0 && (!dst || !src))) return fail(c, MNW_E_ARG, "copy_to_host: bad arguments");
    if (nbytes == 0) return MNW_OK;
    HIPCHK(c, hipSetDevice(c->device));
    return copy_d2h(c, c->stream, dst, src, (size_t)nbytes);
}

int mnw_copy_to_host_batch(mnw_ctx* c, int n, void* const* dst, const void* const* src, const int64_t* nbytes) {
    if (!c || n < 0 || (n > 0 && (!dst || !src || !nbytes))) return fail(c, MNW_E_ARG, "copy_to_host_batch: bad arguments");
    for (int i = 0; i < n; ++i)
        if (nbytes[i] < 0 || (nbytes[i] > 0 && (!dst[i] || !src[i]))) return fail(c, MNW_E_ARG, "copy_to_host_batch: bad arguments");
    HIPCHK(c, hipSetDevice(c->device));
    return copy_d2h_batch(c, c->stream, n, dst, src, nbytes);
}

// ---- GenericCut (scripts/text_to_minh.go:271-312) ------------------------------------------------------
int mnw_generic_cut(mnw_ctx* c, mnw_stream_t stream, const float* mass, float cutoff, int64_t n, const mnw_cut_col* cols, int ncols,
                    int64_t* n_kept) {
    if (!c || !n_kept || n < 0 || ncols < 0 || (ncols > 0 && !cols) || (n > 0 && !mass)) return fail(c, MNW_E_ARG, "cut: bad arguments");
    *n_kept = 0;
    if (n == 0) return MNW_OK;
    HIPCHK(c, hipSetDevice(c->device));
    hipStream_t s = stream ? (hipStream_t)stream : c->stream;
    for (int k = 0; k < ncols; ++k) {
        if (!cols[k].src || !cols[k].dst || (cols[k].elem_bytes != 4 && cols[k].elem_bytes != 8)) return fail(c, MNW_E_ARG, "cut: null column or element size not 4 / 8");
        if (((uintptr_t)cols[k].src | (uintptr_t)cols[k].dst) & (uintptr_t)(cols[k].elem_bytes - 1)) return fail(c, MNW_E_ARG, "cut: misaligned column");
        if (cols[k].src == cols[k].dst) return fail(c, MNW_E_ARG, "cut: a column cannot be filtered in place");
    }
    const int64_t tiles = (n + mnw::kCutTile - 1) / mnw::kCutTile;
    c->text_res = nullptr;                                          // the tile scratch is shared with the text passes
    int rc = ensure_dev(c, &c->d_tile, &c->cap_tile, (size_t)(tiles + 1) * 8 + 16);
    if (rc) return rc;
    {
        ScopedTimer t(c, s, MNW_K_CUT);
        mnw::launch_cut_count(s, mass, cutoff, n, (int64_t*)c->d_tile);
    }
    for (int k0 = 0; k0 < ncols; k0 += mnw::kCutMaxCols) {
        mnw::CutCols cc;
        memset(&cc, 0, sizeof cc);
        cc.n = ncols - k0 < mnw::kCutMaxCols ? ncols - k0 : mnw::kCutMaxCols;
        for (int k = 0; k < cc.n; ++k) { cc.src[k] = cols[k0 + k].src; cc.dst[k] = cols[k0 + k].dst; cc.esz[k] = (uint8_t)cols[k0 + k].elem_bytes; }
        ScopedTimer t(c, s, MNW_K_CUT);
        mnw::launch_cut_scatter(s, mass, cutoff, n, (const int64_t*)c->d_tile, cc);
    }
    HIPCHK(c, hipGetLastError());
    int64_t kept = 0;
    HIPCHK(c, hipMemcpyAsync(&kept, (const int64_t*)c->d_tile + tiles, 8, hipMemcpyDeviceToHost, s));
    HIPCHK(c, hipStreamSynchronize(s));
    *n_kept = kept;
    return MNW_OK;
}

int mnw_text_stats(mnw_ctx* c, int64_t out[3]) {
    if (!c || !out) return MNW_E_ARG;
    for (int k = 0; k < 3; ++k) out[k] = c->text_stats[k];
    return MNW_OK;
}

// ---- raw bit.Array on device buffers ----------------------------------------------------------
int mnw_pack_u64_dev(mnw_ctx* c, mnw_stream_t stream, const uint64_t* x, int64_t n, int bits, uint8_t* out, int64_t cap) {
    if (!c || n < 0 || bits < 0) return fail(c, MNW_E_ARG, "pack: bad arguments");
    if (bits > 64) return fail(c, MNW_E_ARG, "Cannot pack more than 64 bits per element into a bit.Array");  // bit.go:85-87
    if (n == 0 || bits == 0) return MNW_OK;
    if (!x || !out || ((uintptr_t)out & 15)) return fail(c, MNW_E_ARG, "pack: null or misaligned pointer");
    if (cap < mnw::array_bytes(bits, n)) return fail(c, MNW_E_BUFLEN, "pack: output shorter than ArrayBytes(bits, n)");
    HIPCHK(c, hipSetDevice(c->device));
    hipStream_t s = stream ? (hipStream_t)stream : c->stream;
    JobSlot* sl;
    int rc = acquire_slot(c, 1, &sl);
    if (rc) return rc;
    DevJob& d = sl->h_jobs[0];
    memset(&d, 0, sizeof d);
    d.src = x; d.dst = out; d.n = n; d.cap = cap; d.kind = MNW_INT_GROUP;
    d.mode = mnw::kModeExplicit; d.min = 0; d.bits = bits; d.tile0 = 0;
    rc = prepare_slot(c, sl, s, 1, false);
    if (rc) return rc;
    const int64_t tiles = (n + mnw::kTile - 1) / mnw::kTile;
    { ScopedTimer t(c, s, MNW_K_PACK_I64); mnw::launch_pack_i64(s, grid_for(c, tiles), sl->d_jobs, 1, tiles, sl->d_stats, sl->d_meta); }
    HIPCHK(c, hipGetLastError());
    return release_slot(c, sl, s);
}

int mnw_unpack_u64_dev(mnw_ctx* c, mnw_stream_t stream, const uint8_t* in, int64_t in_len, int bits, int64_t n, uint64_t* out) {
    if (!c || n < 0 || bits < 0 || bits > 64) return fail(c, MNW_E_ARG, "unpack: bad arguments");
    if (n == 0) return MNW_OK;
    if (!out) return fail(c, MNW_E_ARG, "unpack: null pointer");
    mnw_dec_job j;
    memset(&j, 0, sizeof j);
    j.src = in; j.dst = out; j.n = n; j.src_len = in_len; j.min = 0; j.bits = bits; j.kind = MNW_INT_GROUP;
    return mnw_decode_blocks(c, stream, &j, 1);
}

// ---- host-pointer synchronous forms (cgo surface) ----------------------------------------------
static int stage_in(mnw_ctx* c, const void* host, size_t bytes) {
    int rc = ensure_dev(c, &c->d_in, &c->cap_in, bytes + 16);
    if (rc) return rc;
    return copy_h2d(c, c->stream, c->d_in, host, bytes);
}

int mnw_pack_u64(mnw_ctx* c, const uint64_t* x, int64_t n, int bits, uint8_t* out, int64_t out_len) {
    if (!c) return MNW_E_ARG;
    if (n < 0 || bits < 0) return fail(c, MNW_E_ARG, "pack: bad arguments");
    if (bits > 64) return fail(c, MNW_E_ARG, "Cannot pack more than 64 bits per element into a bit.Array");
    if (out_len != mnw::array_bytes(bits, n)) return fail(c, MNW_E_BUFLEN, "bit.BufferedArray given buffer of the wrong length");  // bit.go:88-92
    if (out_len == 0) return MNW_OK;
    if (!x || !out) return fail(c, MNW_E_ARG, "pack: null pointer");
    HIPCHK(c, hipSetDevice(c->device));
    int rc = stage_in(c, x, (size_t)n * 8);
    if (rc) return rc;
    rc = ensure_dev(c, &c->d_out, &c->cap_out, (size_t)out_len + 16);
    if (rc) return rc;
    rc = mnw_pack_u64_dev(c, c->stream, (const uint64_t*)c->d_in, n, bits, (uint8_t*)c->d_out, (int64_t)c->cap_out);
    if (rc) return rc;
    return copy_d2h(c, c->stream, out, c->d_out, (size_t)out_len);
}

// bit.ArrayBuffer.Bits (bit.go:151-159): PrecisionNeeded(max(x)), 0 when x is empty.  The max scan runs
// on the GPU (k_reduce_i64 in unsigned mode).
int mnw_bits_u64(mnw_ctx* c, const uint64_t* x, int64_t n, int64_t* bits) {
    if (!c) return MNW_E_ARG;
    if (n < 0 || !bits) return fail(c, MNW_E_ARG, "bits: bad arguments");
    *bits = 0;
    if (n == 0) return MNW_OK;
    if (!x) return fail(c, MNW_E_ARG, "bits: null pointer");
    HIPCHK(c, hipSetDevice(c->device));
    int rc = stage_in(c, x, (size_t)n * 8);
    if (rc) return rc;
    JobSlot* sl;
    rc = acquire_slot(c, 1, &sl);
    if (rc) return rc;
    DevJob& d = sl->h_jobs[0];
    memset(&d, 0, sizeof d);
    d.src = c->d_in; d.n = n; d.kind = MNW_INT_GROUP; d.mode = mnw::kModeUnsigned; d.tile0 = 0;
    rc = prepare_slot(c, sl, c->stream, 1, true);
    if (rc) return rc;
    const int64_t tiles = (n + mnw::kTile - 1) / mnw::kTile;
    { ScopedTimer t(c, c->stream, MNW_K_REDUCE_I64); mnw::launch_reduce_i64(c->stream, grid_for(c, tiles), sl->d_jobs, 1, tiles, sl->d_stats); }
    HIPCHK(c, hipGetLastError());
    uint64_t st[mnw::kStatSlots * mnw::kStatWords];
    HIPCHK(c, hipMemcpyAsync(st, sl->d_stats, sizeof st, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    uint64_t mx = 0;
    for (int k = 0; k < mnw::kStatSlots; ++k) if (st[k * mnw::kStatWords] > mx) mx = st[k * mnw::kStatWords];
    // stat word 0 = bias(max of x ^ signbit) = the unsigned max itself
    *bits = mnw_precision_needed(mx);
    return MNW_OK;
}

int mnw_unpack_u64(mnw_ctx* c, const uint8_t* in, int64_t in_len, int bits, int64_t n, uint64_t* out) {
    if (!c) return MNW_E_ARG;
    if (n < 0 || bits < 0 || bits > 64 || in_len < 0) return fail(c, MNW_E_ARG, "unpack: bad arguments");
    if (n == 0) return MNW_OK;
    if (!out || (bits > 0 && !in)) return fail(c, MNW_E_ARG, "unpack: null pointer");
    if (in_len < mnw::array_bytes(bits, n)) return fail(c, MNW_E_BUFLEN, "unpack: stream shorter than ArrayBytes(bits, n)");
    HIPCHK(c, hipSetDevice(c->device));
    const size_t nb = (size_t)mnw::array_bytes(bits, n);
    int rc = stage_in(c, in, nb);
    if (rc) return rc;
    rc = ensure_dev(c, &c->d_out, &c->cap_out, (size_t)n * 8);
    if (rc) return rc;
    rc = mnw_unpack_u64_dev(c, c->stream, (const uint8_t*)c->d_in, (int64_t)nb, bits, n, (uint64_t*)c->d_out);
    if (rc) return rc;
    return copy_d2h(c, c->stream, out, c->d_out, (size_t)n * 8);
}

static int encode_host(mnw_ctx* c, int kind, const void* x, int64_t n, float low, float high, int64_t pixels,
                       int periodic, int64_t* min, int64_t* bits, uint8_t* out, int64_t cap, int64_t* out_len) {
    if (!c) return MNW_E_ARG;
    if (n < 0 || cap < 0 || !min || !bits || !out_len) return fail(c, MNW_E_ARG, "encode: bad arguments");
    *min = 0; *bits = 0; *out_len = 0;
    if (n == 0) return MNW_OK;
    if (!x) return fail(c, MNW_E_ARG, "encode: null pointer");
    HIPCHK(c, hipSetDevice(c->device));
    const size_t esz = kind == MNW_FLOAT_GROUP ? 4 : 8;
    int rc = stage_in(c, x, (size_t)n * esz);
    if (rc) return rc;
    const int64_t worst = 8 * n;   // 64 bits per element
    const int64_t dcap = cap < worst ? cap : worst;
    rc = ensure_dev(c, &c->d_out, &c->cap_out, (size_t)dcap + 16);
    if (rc) return rc;
    mnw_enc_job j;
    memset(&j, 0, sizeof j);
    j.src = c->d_in; j.dst = c->d_out; j.n = n; j.cap = dcap; j.pixels = pixels; j.low = low; j.high = high;
    j.kind = kind; j.periodic = periodic;
    mnw_block_meta m = {0, 0, 0, 0};          // (a call that fails before the metadata exists leaves zeros, not stack bytes)
    rc = mnw_encode_blocks(c, c->stream, &j, 1, &m);
    *min = m.min; *bits = m.bits;
    if (rc) return rc;
    *out_len = m.nbytes;
    if (m.nbytes > 0) {
        if (!out) return fail(c, MNW_E_ARG, "encode: null output");
        return copy_d2h(c, c->stream, out, c->d_out, (size_t)m.nbytes);
    }
    return MNW_OK;
}

int mnw_int_encode(mnw_ctx* c, const int64_t* x, int64_t n, int64_t* min, int64_t* bits, uint8_t* out, int64_t cap, int64_t* out_len) {
    return encode_host(c, MNW_INT_GROUP, x, n, 0, 0, 0, 0, min, bits, out, cap, out_len);
}

int mnw_float_encode(mnw_ctx* c, const float* x, int64_t n, float low, float high, int64_t pixels, int periodic,
                     int64_t* min, int64_t* bits, uint8_t* out, int64_t cap, int64_t* out_len) {
    return encode_host(c, MNW_FLOAT_GROUP, x, n, low, high, pixels, periodic, min, bits, out, cap, out_len);
}

static int decode_host(mnw_ctx* c, int kind, const uint8_t* in, int64_t in_len, int64_t min, int64_t bits, float low,
                       float high, int64_t pixels, int periodic, uint64_t seed, uint64_t block, int64_t n, void* out) {
    if (!c) return MNW_E_ARG;
    if (n < 0 || bits < 0 || bits > 64 || in_len < 0) return fail(c, MNW_E_ARG, "decode: bad arguments");
    if (n == 0) return MNW_OK;
    if (!out || (bits > 0 && !in)) return fail(c, MNW_E_ARG, "decode: null pointer");
    const int64_t nb = bits == 0 ? 0 : mnw::array_bytes(bits, n);
    if (in_len < nb) return fail(c, MNW_E_BUFLEN, "decode: stream shorter than ArrayBytes(bits, n)");
    HIPCHK(c, hipSetDevice(c->device));
    int rc = stage_in(c, in, (size_t)nb);
    if (rc) return rc;
    const size_t esz = kind == MNW_FLOAT_GROUP ? 4 : 8;
    rc = ensure_dev(c, &c->d_out, &c->cap_out, (size_t)n * esz);
    if (rc) return rc;
    mnw_dec_job j;
    memset(&j, 0, sizeof j);
    j.src = c->d_in; j.dst = c->d_out; j.n = n; j.src_len = nb; j.min = min; j.bits = bits; j.pixels = pixels;
    j.low = low; j.high = high; j.kind = kind; j.periodic = periodic; j.seed = seed; j.block = block;
    rc = mnw_decode_blocks(c, c->stream, &j, 1);
    if (rc) return rc;
    return copy_d2h(c, c->stream, out, c->d_out, (size_t)n * esz);
}

int mnw_int_decode(mnw_ctx* c, const uint8_t* in, int64_t in_len, int64_t min, int64_t bits, int64_t n, int64_t* out) {
    return decode_host(c, MNW_INT_GROUP, in, in_len, min, bits, 0, 0, 0, 0, 0, 0, n, out);
}

int mnw_float_decode(mnw_ctx* c, const uint8_t* in, int64_t in_len, int64_t min, int64_t bits, float low, float high,
                     int64_t pixels, int periodic, uint64_t seed, uint64_t block, int64_t n, float* out) {
    return decode_host(c, MNW_FLOAT_GROUP, in, in_len, min, bits, low, high, pixels, periodic, seed, block, n, out);
}

// ---- minp vectors, host-pointer forms (what the cgo shim over Go slices binds) -----------------------
// vec: [n_file^3][3] float32 on the host.  out: 3*sub_cells^3 slots of slot_bytes each on the host (block b at
// out + b*slot_bytes, meta[b].nbytes bytes valid).  Crosses the host link twice; see DESIGN.md section 6.
int mnw_vectors_encode_host(mnw_ctx* c, const float* vec, int64_t n_file, int64_t sub_cells, const float low[3], const float high[3],
                            const int64_t pixels[3], uint8_t* out, int64_t slot_bytes, mnw_block_meta* meta) {
    int rc = vectors_check(c, n_file, sub_cells, "vectors: n_file must be a positive multiple of sub_cells");
    if (rc) return rc;
    if (!vec || !out || !meta || slot_bytes < 0 || (slot_bytes & 15)) return fail(c, MNW_E_ARG, "vectors: null pointer or slot_bytes not a multiple of 16");
    const int64_t n = n_file * n_file * n_file, nb = 3 * sub_cells * sub_cells * sub_cells;
    HIPCHK(c, hipSetDevice(c->device));
    rc = stage_in(c, vec, (size_t)n * 12);
    if (rc) return rc;
    rc = ensure_dev(c, &c->d_out, &c->cap_out, (size_t)nb * (size_t)slot_bytes + 16);
    if (rc) return rc;
    rc = mnw_vectors_encode(c, c->stream, (const float*)c->d_in, n_file, sub_cells, low, high, pixels, (uint8_t*)c->d_out, slot_bytes, meta);
    if (rc) return rc;
    for (int64_t b = 0; b < nb; ++b)
        if (meta[b].nbytes > 0)
            HIPCHK(c, hipMemcpyAsync(out + b * slot_bytes, (const uint8_t*)c->d_out + b * slot_bytes, (size_t)meta[b].nbytes,
                                     hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return MNW_OK;
}

int mnw_vectors_decode_host(mnw_ctx* c, const uint8_t* in, int64_t slot_bytes, const mnw_block_meta* meta, int64_t n_file,
                            int64_t sub_cells, const float low[3], const float high[3], const int64_t pixels[3], int periodic, float L,
                            uint64_t seed, uint64_t first_block, float* vec_out) {
    int rc = vectors_check(c, n_file, sub_cells, "vectors: n_file must be a positive multiple of sub_cells");
    if (rc) return rc;
    if (!in || !vec_out || !meta || slot_bytes < 0 || (slot_bytes & 15)) return fail(c, MNW_E_ARG, "vectors: null pointer or slot_bytes not a multiple of 16");
    const int64_t n = n_file * n_file * n_file, nb = 3 * sub_cells * sub_cells * sub_cells;
    HIPCHK(c, hipSetDevice(c->device));
    rc = ensure_dev(c, &c->d_in, &c->cap_in, (size_t)nb * (size_t)slot_bytes + 16);
    if (rc) return rc;
    for (int64_t b = 0; b < nb; ++b)
        if (meta[b].nbytes > 0)
            HIPCHK(c, hipMemcpyAsync((uint8_t*)c->d_in + b * slot_bytes, in + b * slot_bytes, (size_t)meta[b].nbytes, hipMemcpyHostToDevice, c->stream));
    rc = ensure_dev(c, &c->d_out, &c->cap_out, (size_t)n * 12);
    if (rc) return rc;
    rc = mnw_vectors_decode(c, c->stream, (const uint8_t*)c->d_in, slot_bytes, meta, n_file, sub_cells, low, high, pixels, periodic, L, seed,
                            first_block, (float*)c->d_out);
    if (rc) return rc;
    return copy_d2h(c, c->stream, vec_out, c->d_out, (size_t)n * 12);
}

}  // extern "C"
