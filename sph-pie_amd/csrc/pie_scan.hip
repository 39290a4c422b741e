// pie_scan.hip — C ABI (include/pie_scan.h) over the HIP kernels of pie_kernels.h.  gfx950 only.
// There is no CPU path in this library: every entry point either runs on the GPU or returns an error.
//
// One stream, two slots.  A scan is one bandwidth-bound kernel (K1, the table pass) followed by short latency-bound
// work (K2: offsets + the order of every small bucket; K3 scatter / K4 order only for buckets that outgrew their 16
// direct slots), all on ONE stream: co-running the tail with the next table pass on other streams was measured and
// rejected (its dependent loads crawl under a saturated HBM and every cross-queue dependency costs ~15 us:
// profiles/r01_d_two_stream_timeline.txt).
// The big per-scan buffers exist twice (two SLOTS) and the small histogram "spans" three times, rotating: K2 of
// every scan zeroes the span the scan AFTER THE NEXT will use (its last user is two launches old), so no memset and no
// event sits between kernels — a marker packet between two kernels costs ~6 us here, a plain kernel boundary ~0.
// pie_scan_begin enqueues K1+K2 and returns; pie_scan_finish spins on the summary that K2's last block writes to
// mapped host memory, then enqueues K3/K4 sized from it if any bucket needs them.  With begin(i+1) called before
// finish(i) the stream always holds the next table pass, so the host's round trip is off the critical path:
//   stream:  K1(i) | K1(i+1)+K2(i) in one launch | [K3(i) K4(i)] consumers(i) | K1(i+2)+K2(i+1) | ...
// (K2 of a scan rides in the first blocks of the next scan's table-pass launch; a scan nobody follows gets its K2 from
// pie_scan_finish).
// The table carries three derived columns (2-byte and 1-byte liveness keys, 16-byte payload records; pie_kernels.h)
// that every writer of `end` keeps in step; the scan form is chosen per scan from what the previous scan observed
// (live fraction, ambiguous keys, hot buckets) — see scan_begin.
#include "../../include/pie_scan.h"
#include "pie_kernels.h"
#include "pie_ordered.h"

#include <hip/hip_runtime.h>

#include <cstring>
#include <rocprim/device/device_radix_sort.hpp> // the archive chain's two sorts and its scan (pie_archive_queue)
#include <rocprim/device/device_scan.hpp>

#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <algorithm>
#include <cstring>
#include <ctime>
#include <string>
#include <vector>

#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

using namespace pie;

namespace {

thread_local char g_create_error[256] = "";

struct ScanEvents {
    hipEvent_t e0, e1, e2; // before K1, after K1, after the last kernel of the tail
};

constexpr int kEventRing = 2048;
constexpr size_t kDirectMaxBytes = (size_t)2 << 30; // direct bucket slots (256 B per user) are carried up to 8 M users

// everything one scan owns
struct Slot {
    int* counts = nullptr;         // the span this scan uses: histogram (transposed user order, see hist_index) | tile_pub[] | ScanCtl | Summary
    int* counts_ord = nullptr;     // per-user counts in user order, written by K2: what every consumer after K2 reads
    unsigned long long* tile_pub = nullptr;
    ScanCtl* ctl = nullptr;
    Summary* sum = nullptr;
    HostSummary* h_sum = nullptr;     // mapped pinned memory: written by K2's last block
    HostSummary* h_sum_dev = nullptr; // the same memory as the device sees it
    unsigned long long seq = 0;       // value h_sum->seq takes when this slot's scan has its summary
    long long* offsets = nullptr;
    SelRec* sel = nullptr;
    int* sel_rank = nullptr;
    int* blk_count = nullptr;
    BktRec* bkt = nullptr;
    BktRec* direct = nullptr;      // kTinyMax direct bucket slots per user (nullptr: user table too large, staged route only)
    int* blk_hot_base = nullptr;   // per (K1 block, hot user): the base its block-level histogram atomic returned
    int* hot_list = nullptr;       // K2's output: users whose bucket exceeded the hot threshold (<= kHotMax kept)
    int* over_list = nullptr;      // K2's output: buckets with staged records (user id, or -1 when nothing sits in its direct slots)
    HotSet hot{};                  // hot users of THIS scan (K1 and K3 must agree)
    bool k2_pending = false;       // K1 is queued, K2 is not yet: it rides in the next scan's launch or is launched by finish
    int4* zero_span = nullptr;     // the histogram span this scan's K2 zeroes (the one the scan after the next will use)
    int* msg = nullptr;            // this scan's result message (caller-owned device memory), or nullptr
    int msg_u_pad = 0;
    long long msg_cap = 0;
    int* msg_counts = nullptr;     // optional second destination: the per-user counts (pie_scan_begin_packed2)
    bool msg_by_k2 = false;        // K2 wrote the message (fused route)
    int* out_idx = nullptr;
    Segment* seg_list = nullptr;
    Segment* small_list = nullptr;
    int* part_cursor = nullptr;   // in the span: cursors of the partitioned fast path
    SelRec* part_rec = nullptr;   // kPartMax x kPartCap records
    bool fast = false;            // this scan ran on the partitioned fast path
    bool ordered = false;         // this scan ran on the ordered run (pie_ordered.h): nothing left to launch at finish
    long long q_now = 0, q_cutoff = 0; // the query, kept for a rerun on the general path
    int* big_list = nullptr;
    // per-scan host state
    bool in_flight = false;
    bool have_result = false;
    Summary last{};
    int k1_blocks = 0;
    long long rows_per_block = 0;
    int variant = 0;
    int ev_index = -1; // index into the event ring, -1 = not profiled
};

// Batches in flight: three.  With two, begin(i + 2) has to wait for finish(i), whose summary arrives half-way through launch i + 1
// (the tail of batch i rides there): the host then has the second half of ONE launch to prepare and queue the next, and every
// microsecond it is late the GPU idles (8 - 12 us per 54 us launch at 64 queries per batch, rocprofv3 timeline).  With three
// the next launch is always queued before the host waits.
constexpr int kBatchSlots = 3;
constexpr int kLaneMax = 4;     // batch lanes of a context (pie_set_batch_lanes): independent streams, kBatchSlots batches in flight on each

// everything one batch of queries owns (see pie_kernels.h "batched scan").  The primary result of a batch on the general pass
// is the UNION (uoff / urows / umlo / umhi); per-query lists are materialised from it on request, or produced directly by the
// paths that work per query (fallback scans on the general path, a batch on the ordered run).
struct BatchSlot {
    int n_q = 0;
    bool in_flight = false, k2_pending = false, have_result = false;
    int dshift = 4;                    // union bucket capacity (log2) the batch ran with
    bool ordered = false;              // this batch ran on the ordered run (pie_ordered.h "batched form"): no tail, nothing rides
    bool unsupported = false;          // this table cannot run the batched pass (no key columns / no direct slots): every query falls back
    bool fine_key = false;
    unsigned long long seq = 0;
    char* span = nullptr;              // this batch's span (histogram | tile granules | ctl | summary + row statistics | mq slots)
    char* zero_span = nullptr;         // the span its tail zeroes
    BktRec* direct = nullptr;          // union bucket slots: [cap_users << bdshift]; BktRec::pad = queries 0..31 that selected the row
    unsigned* direct_hi = nullptr;     // ... queries 32..63
    long long* uoff = nullptr;         // the union result: [cap_users + 1]
    int* urows = nullptr;              // [cap_users << bdshift]
    unsigned* umlo = nullptr;
    unsigned* umhi = nullptr;
    bool union_ok = false;             // the union arrays hold every query of the finished batch
    bool union_part = false;           // ... the queries that did not fall back
    unsigned long long mu = 0;         // union rows
    BatchHost* bh = nullptr;           // mapped pinned: the tail's summary + per-query totals
    BatchHost* bh_dev = nullptr;
    int lists_q = 0;                   // queries the per-query list storage below holds (0: not allocated)
    int* counts_ord = nullptr;         // [lists_q][users_stride]
    long long* offsets = nullptr;      // [lists_q][users_stride]
    int* out_idx = nullptr;            // [lists_q][out_stride]
    bool list_ok[kBatchMax];           // query q's counts / offsets / row list are in the list storage
    pie_query q[kBatchMax];
    bool fallback[kBatchMax];          // rerun on the general path (dense query, outgrown bucket, bad rows)
    Summary last[kBatchMax];
    int* over_idx[kBatchMax] = {};     // row list of a query that outgrew out_stride
    long long over_cap[kBatchMax] = {};
    int* idx_of[kBatchMax] = {};       // where query q's row list lives (valid with list_ok[q])
    int msg_kind = 0;                  // 0: none; 1: one message per query (lists); 2: ONE union message
    int* msg = nullptr;                // caller-owned device-visible memory
    long long msg_stride = 0, msg_cap = 0;
    int msg_u_pad = 0;
    int* msg_counts = nullptr;
    long long msg_counts_stride = 0;
    int k1_blocks = 0;
    int ev_index = -1;
    int lane = 0;                      // the lane (stream) the batch ran on
    hipStream_t stream = nullptr;
    bool main_ordered = true;          // the context's main stream is ordered behind the batch's kernels (see order_after_batch)
};

// the ordered run of the resident table (pie_ordered.h)
constexpr int kOrdAppendMax = 4096; // rows of one append the run takes in place (a larger append drops the run)

struct OrderedRun {
    int grid_mult = 12;      // PIE_ORD_GRID: blocks per CU of the key-stream kernels on the 1-byte key
    int mode = 1;            // PIE_ORDERED: 0 = never, 1 = when the general path is weak (dense / skewed queries), 2 = always
    bool valid = false;
    long long n = 0;         // positions (= rows the all-selecting scan returned)
    long long rows = 0;      // table rows the run covers (built from, plus the appends it took)
    long long held = 0;      // rows it holds (the selectable ones of those)
    long long cap = 0;       // positions the arrays hold
    int cap_users = 0;
    OrdRec* pay = nullptr;
    long long* end = nullptr;
    lkey_t* key = nullptr;
    fkey_t* fkey = nullptr;
    int* pos = nullptr;
    long long* uoff = nullptr;               // [users + 1] first position of every user's segment (rows, then spare slots)
    int* ufill = nullptr;                    // [users] rows in the segment
    int* pend = nullptr;                     // [users] rows of an append that found their segment full (re-spread)
    int* bhead = nullptr;                    // [users] head of the user's chain of rows in the append in progress (-1 between appends)
    int* bnext = nullptr;                    // [kOrdAppendMax] the chains: the row of the same user that came before this one, or -1
    int* placed = nullptr;                   // [2][4096] outcome per row of the append in progress (+ the copy pass 2 reads)
    OrdRec* alt_pay = nullptr;               // the arrays a re-spread moves the run into (allocated at the first one), then swapped
    long long* alt_end = nullptr;
    lkey_t* alt_key = nullptr;
    fkey_t* alt_fkey = nullptr;
    long long* alt_uoff = nullptr;
    unsigned long long respreads = 0;
    // batched form: per query unit counts / prefixes / group sums, one summary set per batch slot
    int* bq_local = nullptr;
    long long* bq_gsum = nullptr;
    long long* bq_gbase = nullptr;
    OrdCtl* bq_ctl = nullptr;
    char* bq_sum[2] = {nullptr, nullptr};
    int users = 0;                           // users that have a segment (>= n_users: room for users yet to come)
    long long pos_cap = 0;                   // positions the arrays hold
    int* unit_count[2] = {nullptr, nullptr}; // alternate: the finish kernel of one ordered scan zeroes the other buffer
    int uc_next = 0;
    long long units_cap = 0;
    int* unit_local = nullptr;               // exclusive prefix of the unit counts inside their group of 1024
    long long* group_sum = nullptr;
    long long* group_base = nullptr;
    unsigned long long* tile_ballot = nullptr;
    unsigned int* tile_prefix = nullptr;
    unsigned int* h_stale = nullptr;         // mapped pinned: rows outside the run that a writer of `end` brought back to life
    unsigned int* stale = nullptr;           // ... as the device sees it
    bool no_room = false;                    // the arrays did not fit: not tried again for this table
    char* sum[2] = {nullptr, nullptr};       // per scan slot: Summary + row-statistics slots + OrdCtl
    unsigned wanted = 0;     // consecutive scans that wanted the run while it was not there
    unsigned need = 2;       // ... and how many of them it takes to build it: 2, more after a run that did not live long
    unsigned long long built_at = 0; // scans_begun when it was last built
    unsigned long long builds = 0;
    double build_ms = 0;
};

} // namespace

struct pie_ctx {
    int device = -1;
    int n_cus = 0;
    hipStream_t own_stream = nullptr;
    hipStream_t stream = nullptr; // main stream (own_stream or the caller's)
    unsigned long long seq_counter = 0;
    char err[512] = "";

    // resident table
    long long n = 0;
    int n_users = 0;
    long long cap_rows = 0;
    long long sel_cap = 0;      // records the staging arrays (sel, sel_rank) of each slot hold: see ensure_sel
    int cap_users = 0;
    long long *d_start = nullptr, *d_end = nullptr;
    int *d_user = nullptr, *d_disc = nullptr;
    // derived liveness-key column (see pie_kernels.h): 2 B/row, kept in step with d_end by every writer
    lkey_t* d_key = nullptr;
    PayRec* d_pay = nullptr;    // derived payload column (start, user, disc) per row: immutable after a row is written
    fkey_t* d_fkey = nullptr;   // 7-bit key of the top of the `end` range (1 B/row), see pie_kernels.h
    long long fkey_base = 0;
    int fkey_shift = 0;
    bool fkey_poor = false;     // a fine-keyed scan found too many ambiguous rows: use the 15-bit key
    unsigned int* d_hist = nullptr;
    std::vector<unsigned int> key_hist;  // host copy of the key histogram of the last full key build (8 keys per bin)
    std::vector<unsigned long long> key_hist_above; // key_hist_above[b] = rows in bins b .. end (what every begin asks for, per query)
    long long key_hist_rows = 0;         // rows it covers
    long long key_base = 0;
    int key_shift = 0;
    bool key_ok = false;        // d_key covers rows [0, n) under (key_base, key_shift)
    bool key_dirty = false;     // rows were appended / re-ended since the column was built: a rebuild may fit better
    bool key_poor = false;      // a keyed scan found too many ambiguous rows and a rebuild would not help: use the `end` column
    bool key_rebuild = false;   // ... and a rebuild may help: done at the next pie_scan_begin with nothing in flight
    bool keyed_enabled = true;  // PIE_K1_KEYED=0 turns the keyed form off
    int order_block = 512;       // threads (= users) per block of the fused K2 + order kernel (PIE_ORDER_BLOCK: 256 / 512 / 1024)
    bool no_ride = false;        // PIE_K2_RIDE=0: K2 never rides in the next scan's launch (A/B runs)
    bool no_fused_order = false; // PIE_FUSED_ORDER=0: K2 and the tiny-bucket order as two kernels (A/B runs)
    int k1_keyed = 0xC85;       // keyed liveness-first form (bit 0x400; 0x800: the 1-byte fine key where the query allows), unroll 8
    long long* d_range = nullptr;
    char* h_stage = nullptr;       // pinned host + device staging of the small mutations (append, touch): grown, never per call
    char* d_stage = nullptr;
    size_t stage_bytes = 0;
    bool expired_copy_total = false; // PIE_EXPIRED_COPY_TOTAL=1: the queue's length by a copy behind the gather, as before round 3 (A/B runs)
    // Appends and touches that need nothing back from the device (no ordered run to keep in step: its spare-slot bookkeeping is
    // read by the host) do not wait for it: the rows are validated on the host, staged in one of two pinned areas and queued; the
    // next scan runs behind them in stream order.  A server's turn — logins, touches, one feed scan — is then ONE wait, not three.
    struct AsyncStage { char* h = nullptr; char* d = nullptr; size_t bytes = 0; hipEvent_t ev = nullptr; bool pending = false; } astage[2];
    int astage_next = 0;
    bool async_mutations = true;   // PIE_ASYNC_MUTATIONS=0: every append / touch waits for its kernel (A/B runs)
    int* d_shard_rows = nullptr;   // pie_shard_table: local row -> global row
    int* d_shard_users = nullptr;  // ... local user -> global user
    long long shard_rows_n = 0;
    int shard_users_n = 0;

    // predicate table
    unsigned long long disc_mask = ~0ull;
    int n_disc = 64;

    // scan plans: [0] streaming form, [1] liveness-first form, [2] keyed liveness-first form, [3] the same on the 1-byte key
    int plan_blocks[4] = {0, 0, 0, 0};
    long long plan_rows[4] = {0, 0, 0, 0};
    int n_tiles = 0;
    int k1_variant = 0x03;    // streaming form: nontemporal loads + late user materialisation
    int k1_live_first = 0x85; // liveness-first form (unroll 8, nontemporal), chosen when few rows are live
    bool k1_pinned = false;   // PIE_K1_VARIANT given: no adaptation
    double live_frac = -1;    // live fraction seen by the last finished scan of this table (-1: none yet)
    int dshift = 4;           // log2 of the direct-slot capacity per user (16 .. kSmallMax); grown when scans outgrow it
    int dshift_want = 4;      // capacity the last finished scan asked for (applied at the next pie_scan_begin with nothing in flight)
    bool hot_bucket = false;  // the last finished scan had one bucket with > 1/64 of the selected rows
    bool clustered = false;   // the last streaming scan found most selected rows next to a row of the same user
    HotSet hot{};             // users whose buckets were "big" in a recent scan: block-level histogram in the aggregated forms
    unsigned hot_seen = 0;    // size of the hot list the set was read from
    unsigned hot_age = 0;     // finished scans since the set was read
    long long last_m = -1;    // M of the last finished feed scan of this table (-1: none yet)
    int part_shift = -1;      // users per partition = 1 << part_shift (-1: too many users for the fast path)
    int n_parts = 0;
    bool fast_enabled = false; // the partitioned path is opt-in (PIE_FAST_PATH=1): measured at parity with the general
    bool fast_env = false;     // path (0.179 vs 0.177 ms/step), so the simpler path stays the default; an overflow turns it off for the table

    UnionRow* d_union = nullptr;   // pie_batch_pack_union_device: per user kUnionMax union rows
    int* d_union_cnt = nullptr;    // ... their counts, padded with zeros to whole groups of 1024 (the two-level prefix reads groups)
    int* d_union_local = nullptr;  // ... prefix inside the group
    long long* d_union_off = nullptr; // ... group sums | group bases | a Summary the prefix kernel writes M into | the overflow flag
    int union_users = 0;
    OrderedRun ord;
    bool ord_building = false;  // the scan being begun is the ordered run's build
    Slot slot[2];
    // Batch lanes.  A batch over a SHARD-sized table (a tenth of cfg3) is one launch of ~20 us that fills a fraction of the chip:
    // its time is latency (key load -> candidate gather -> atomic -> store; then the tail's prefix chain), not bytes.  Lanes are
    // independent pipelines — own stream, own three slots, own spans — so batches of different lanes run SIDE BY SIDE on the chip;
    // begin deals batches to the lanes round robin, finish returns them in the order they were begun (profiles/r03_zg_lanes.txt:
    // 1.25 x 10^7 rows, 64 queries: 20.9 us per batch on one lane, 7.8 on four).  Lane 0 is the context's main stream.
    BatchSlot bslot[kLaneMax * kBatchSlots]; // lane l: slots [l * kBatchSlots, (l + 1) * kBatchSlots)
    char* bspan[kLaneMax * 3] = {};          // lane l: rotating batch spans [3 l, 3 l + 3) (batch_span_bytes each)
    hipStream_t lane_stream[kLaneMax] = {};  // [0] unused (= stream); the others are created with the lane
    hipEvent_t lane_event[kLaneMax][2] = {}; // per lane: [0] orders the lane behind the main stream (idle_epoch), [1] the main stream behind the lane (order_after_batch)
    int n_lanes = 1;                         // lanes begin deals to (pie_set_batch_lanes / PIE_BATCH_LANES; 0 there = by table size)
    int lanes_want = 0;                      // what the caller asked for (0: automatic)
    int lane_rr = 0;                         // lane the next begin tries first
    int lane_flight[kLaneMax] = {};          // batches in flight per lane
    int lane_next[kLaneMax] = {};            // slot (0..kBatchSlots) the lane's next begin uses
    int lane_span_next[kLaneMax] = {};
    bool lane_alloc[kLaneMax] = {};          // the lane's slot arrays and spans exist
    unsigned char fifo[kLaneMax * kBatchSlots] = {}; // lanes of the batches in flight, in the order they were begun
    int fifo_head = 0;
    // Table changes (loads, appends, touches, key builds) are queued on the MAIN stream and only happen while no batch is in
    // flight; some return with kernels still queued (the fine key of a load).  A lane's stream knows nothing of the main
    // stream, so the first batch a lane begins after the context was idle waits for what the main stream holds.
    unsigned long long idle_epoch = 1;                // bumped whenever the last batch in flight leaves
    unsigned long long lane_epoch[kLaneMax] = {};     // the idle epoch the lane's stream was last ordered behind the main stream in
    long long* d_mat_tile = nullptr;   // materialisation scratch: [kBatchMax][tiles + 1] tile sums / prefixes
    unsigned int* d_mat_qmax = nullptr; // ... [kBatchMax] largest per-user count
    int mat_tiles = 0;
    int b_flight = 0;           // batches begun and not finished, all lanes
    BatchSlot* bres = nullptr;  // last finished batch
    int bdshift = 4;            // log2 of the union bucket capacity of the batched pass (16 .. 64 slots per user)
    int bdshift_want = 4;       // capacity the last finished batch asked for (applied at the next begin with nothing in flight)
    int run_shift = 0;          // chunk interleave of the keyed / batched table pass: 0 = fully interleaved (dense stretches of
                                // live rows, the safe default), 3 = eight consecutive chunks per wave (live rows spread evenly: a few
                                // per cent faster); follows the densest-chunk statistic of the last pass, see choose_run_shift
    bool batch_poor = false;    // union buckets overflowed at their largest capacity (skewed users): batches run as single scans
    bool ord_lists_only = false; // a batch's union on the ordered run outgrew the result arrays: its batches take the per-query chain
    bool run_shift_pinned = false; // PIE_RUN_SHIFT=0..3 pins it (A/B runs)
    bool last_was_batch = false; // pie_stats_get describes the last finished batch rather than the last single scan
    unsigned long long bseq_counter = 0;
    char* span[3] = {nullptr, nullptr, nullptr}; // rotating histogram spans (see counts_span)
    int span_next = 0;
    int profile_every = 1;   // with profiling on, every n-th scan carries timing events
    double index_build_ms = 0;         // pie_table_info: last full build of the derived columns
    double wait_deadline_ms = 20000.0; // bound of every wait on a scan summary (PIE_WAIT_DEADLINE_MS)
    unsigned long long scans_begun = 0;
    int next_slot = 0;   // slot the next pie_scan_begin uses
    int n_flight = 0;    // scans begun and not finished (0..2)
    Slot* res = nullptr; // last finished scan (results)
    const unsigned char* d_qual = nullptr; // group-qualified scan form (k_scan_compact<.., GQ>): per-user flag
    char* d_arch = nullptr;                // pie_archive_queue: per-group scratch
    size_t arch_bytes = 0;
    char* d_arch_tmp = nullptr;            // ... temporary storage of its sorts / scan
    size_t arch_tmp_bytes = 0;
    unsigned long long arch_alg_bytes = 0; // algorithmic bytes of the last archive queue (32 B/row + 4 B per queued row)
    double arch_ms_sum = 0;                // with profiling on: device time of the archive chains (first kernel -> last sort)
    unsigned arch_calls = 0;

    // shared scratch
    long long* d_blk_off = nullptr;
    Summary* d_summary = nullptr; // table-maintenance calls
    Summary* h_summary = nullptr; // pinned

    // profiling
    bool profiling = false;
    std::vector<ScanEvents> ring;
    int ring_used = 0;
    double k1_ms_sum = 0, scan_ms_sum = 0;
    unsigned n_profiled = 0;
};

namespace {

int fail(pie_ctx* c, int code, const char* fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    if (c) vsnprintf(c->err, sizeof c->err, fmt, ap);
    else vsnprintf(g_create_error, sizeof g_create_error, fmt, ap);
    va_end(ap);
    return code;
}

// Host-time profile of the batched begin / finish (builds with -DPIE_HOST_PROF only: tools/host_prof.sh); totals go to stderr
// when a context is destroyed.
#ifdef PIE_HOST_PROF
struct HostProf {
    double acc[16] = {0};
    unsigned long long n[16] = {0};
    timespec last{};
    void start() { clock_gettime(CLOCK_MONOTONIC, &last); }
    void tick(int i)
    {
        timespec t{};
        clock_gettime(CLOCK_MONOTONIC, &t);
        acc[i] += (double)(t.tv_sec - last.tv_sec) * 1e6 + (double)(t.tv_nsec - last.tv_nsec) * 1e-3;
        n[i]++;
        last = t;
    }
    void report() const
    {
        static const char* names[16] = {"begin: checks + dense test", "begin: slot / span set-up", "begin: tables", "begin: launch", "begin: rest",
                                        "finish: k2 launch", "finish: wait", "finish: rest", "", "", "", "", "", "", "", ""};
        for (int i = 0; i < 8; ++i)
            if (n[i]) fprintf(stderr, "[pie host prof] %-28s %8.3f us x %llu\n", names[i], acc[i] / (double)n[i], n[i]);
    }
};
HostProf g_prof;
#define PIE_PROF_START() g_prof.start()
#define PIE_PROF_TICK(i) g_prof.tick(i)
#else
#define PIE_PROF_START() ((void)0)
#define PIE_PROF_TICK(i) ((void)0)
#endif

#define PIE_HIP(c, call)                                                                              \
    do {                                                                                              \
        hipError_t e_ = (call);                                                                       \
        if (e_ != hipSuccess)                                                                         \
            return fail((c), e_ == hipErrorOutOfMemory ? PIE_E_NOMEM : PIE_E_HIP, "%s: %s (%s:%d)", #call, \
                        hipGetErrorString(e_), __FILE__, __LINE__);                                   \
    } while (0)

template <class T>
void dfree(T*& p)
{
    if (p) (void)hipFree(p);
    p = nullptr;
}

DirectSlots direct_of(const pie_ctx* c, const Slot& sl);

// scans in flight are the last n_flight slots handed out; this is the older one
Slot* oldest_in_flight(pie_ctx* c)
{
    if (c->n_flight == 0) return nullptr;
    return &c->slot[c->n_flight == 2 ? c->next_slot : (c->next_slot ^ 1)];
}

void free_slots(pie_ctx* c)
{
    for (Slot& s : c->slot) {
        s.counts = nullptr; s.sum = nullptr;
        s.tile_pub = nullptr; s.ctl = nullptr;
        dfree(s.offsets); dfree(s.counts_ord); dfree(s.sel); dfree(s.sel_rank); dfree(s.blk_count);
        dfree(s.bkt); dfree(s.direct); dfree(s.blk_hot_base); dfree(s.hot_list); dfree(s.over_list); dfree(s.out_idx); dfree(s.seg_list); dfree(s.small_list); dfree(s.big_list); dfree(s.part_rec); s.part_cursor = nullptr;
        s.in_flight = s.have_result = false;
    }
    for (char*& sp : c->span) dfree(sp);
    c->span_next = 0;
    c->sel_cap = 0;
    c->res = nullptr;
    c->n_flight = 0;
    c->next_slot = 0;
}

void free_batch(pie_ctx* c)
{
    for (BatchSlot& b : c->bslot) {
        dfree(b.counts_ord); dfree(b.offsets); dfree(b.direct); dfree(b.out_idx);
        dfree(b.direct_hi); dfree(b.uoff); dfree(b.urows); dfree(b.umlo); dfree(b.umhi);
        b.lists_q = 0;
        for (int q = 0; q < kBatchMax; ++q) { dfree(b.over_idx[q]); b.over_cap[q] = 0; b.idx_of[q] = nullptr; b.list_ok[q] = false; }
        b.in_flight = b.k2_pending = b.have_result = b.union_ok = b.union_part = false;
        b.n_q = 0;
    }
    for (char*& sp : c->bspan) dfree(sp);
    dfree(c->d_mat_tile); dfree(c->d_mat_qmax);
    c->mat_tiles = 0;
    for (int l = 0; l < kLaneMax; ++l) {
        c->lane_flight[l] = c->lane_next[l] = c->lane_span_next[l] = 0;
        c->lane_alloc[l] = false;
    }
    c->fifo_head = 0;
    c->lane_rr = 0;
    c->b_flight = 0;
    c->bres = nullptr;
}

void ord_free(pie_ctx* c)
{
    OrderedRun& o = c->ord;
    dfree(o.pay); dfree(o.end); dfree(o.key); dfree(o.fkey); dfree(o.pos); dfree(o.uoff); dfree(o.ufill); dfree(o.pend); dfree(o.placed); dfree(o.bhead); dfree(o.bnext);
    dfree(o.alt_pay); dfree(o.alt_end); dfree(o.alt_key); dfree(o.alt_fkey); dfree(o.alt_uoff);
    dfree(o.bq_local); dfree(o.bq_gsum); dfree(o.bq_gbase); dfree(o.bq_ctl); dfree(o.bq_sum[0]); dfree(o.bq_sum[1]);
    dfree(o.unit_count[0]); dfree(o.unit_count[1]); dfree(o.unit_local); dfree(o.group_sum); dfree(o.group_base); dfree(o.tile_ballot); dfree(o.tile_prefix);
    dfree(o.sum[0]); dfree(o.sum[1]);
    o.valid = false;
    o.no_room = false;
    o.cap = 0; o.cap_users = 0; o.units_cap = 0; o.n = 0; o.rows = 0; o.wanted = 0; o.need = 2;
}

// The run no longer describes the table.  new_table: a load / re-shard — nothing is known about what comes.  Otherwise a row
// arrived that the run could not take (out of time order, no room even after a re-spread) or one it does not hold came back
// to life: a run that lived for fewer than 64 scans was not worth its build, so the next one takes four times as many
// consecutive dense / skewed queries to come about (2, 8, 32, ... 4096); one that lived longer starts over at 2.
void ord_invalidate(pie_ctx* c, bool new_table = false, bool out_of_order = false)
{
    OrderedRun& o = c->ord;
    if (new_table) {
        o.need = 2;
        o.wanted = 0;
    } else if (o.valid) {
        // rows that arrive out of time order are how this table is fed: the run cannot live on it, one build was enough to learn that
        if (out_of_order && c->scans_begun - o.built_at < 64) o.need = 4096u;
        else o.need = c->scans_begun - o.built_at < 64 ? (o.need * 4 < 4096u ? o.need * 4 : 4096u) : 2u;
        o.wanted = 0;
    }
    o.valid = false;
}

// spare slots every segment gets on top of a sixteenth of its rows: 16 where that costs under a quarter of the rows
// (a user with a handful of rows then takes a few hundred appends' worth of new sessions before anything has to move), else 4
int ord_spare(const pie_ctx* c) { return (long long)16 * c->cap_users <= c->cap_rows / 4 ? 16 : 4; }

OrdMirror ord_mirror_of(const pie_ctx* c)
{
    OrdMirror m{};
    if (c->ord.valid) {
        m.pos = c->ord.pos;
        m.end = c->ord.end;
        m.key = c->ord.key;
        m.fkey = c->ord.fkey;
        m.stale = c->ord.stale;
    }
    return m;
}

void free_table(pie_ctx* c)
{
    for (int l = 1; l < kLaneMax; ++l) // nothing of a lane's may still be running when its arrays go
        if (c->lane_stream[l]) (void)hipStreamSynchronize(c->lane_stream[l]);
    dfree(c->d_union); dfree(c->d_union_cnt); dfree(c->d_union_local); dfree(c->d_union_off);
    c->union_users = 0;
    ord_free(c);
    free_batch(c);
    dfree(c->d_start); dfree(c->d_end); dfree(c->d_user); dfree(c->d_disc); dfree(c->d_key); dfree(c->d_pay); dfree(c->d_fkey);
    dfree(c->d_arch);
    c->arch_bytes = 0;
    dfree(c->d_arch_tmp);
    c->arch_tmp_bytes = 0;
    c->key_ok = false;
    dfree(c->d_blk_off);
    free_slots(c);
    c->cap_rows = 0; c->cap_users = 0; c->n = 0; c->n_users = 0;
}

DirectSlots direct_of(const pie_ctx* c, const Slot& sl)
{
    DirectSlots d;
    d.p = sl.direct;
    d.shift = c->dshift;
    return d;
}

int sync_all(pie_ctx* c)
{
    PIE_HIP(c, hipStreamSynchronize(c->stream));
    for (int l = 1; l < kLaneMax; ++l)
        if (c->lane_stream[l]) PIE_HIP(c, hipStreamSynchronize(c->lane_stream[l]));
    return PIE_OK;
}

// Grid of the scan kernel: every block owns a contiguous, tile-aligned row range.  Two plans, measured on
// cfg3 (profiles/r01_c_*): the streaming form likes many short blocks (short tail), the liveness-first form likes
// fewer, longer blocks (each wave ends with one partial drain of its live-row ring, amortised over more rows).
// PIE_K1_BLOCKS / PIE_K1_BLOCKS_LIVE override for tuning.
void plan_one(pie_ctx* c, int which, long long want, const char* env)
{
    if (const char* e = getenv(env)) {
        long long v = atoll(e);
        if (v > 0) want = v;
    }
    // a whole number of wave-tiles for every unroll (the keyed form reads 512 rows per wave per load)
    const long long block_tile_rows = which == 3 ? (long long)kFineKeyRowsPerLoad * 8 * kK1Waves
                                     : which == 2 ? (long long)kKeyRowsPerLoad * 8 * kK1Waves : (long long)kUnitRows * 8 * kK1Waves;
    long long tiles = (c->n + block_tile_rows - 1) / block_tile_rows;
    if (tiles < 1) tiles = 1;
    if (want > tiles) want = tiles;
    if (want < 1) want = 1;
    const long long tiles_per_block = (tiles + want - 1) / want;
    c->plan_rows[which] = tiles_per_block * block_tile_rows;
    c->plan_blocks[which] = (int)((c->n + c->plan_rows[which] - 1) / c->plan_rows[which]);
    if (c->plan_blocks[which] < 1) c->plan_blocks[which] = 1;
}

// Lanes of the batched scan (see pie_ctx::bslot): as many as the caller pinned, else by the size of the table — a batch over a
// small table is latency, and several side by side fill the chip; over the whole of cfg3 one batch already does
// (profiles/r03_zg_lanes.txt).
void choose_lanes(pie_ctx* c)
{
    if (c->lanes_want > 0) c->n_lanes = c->lanes_want;
    else c->n_lanes = c->n <= (1LL << 25) ? 4 : 3; // (10^8 rows: 53 / 42.5 / 41.5 / 41.6 us per 64-query batch on 1 / 2 / 3 / 4 lanes)
}

void plan_k1(pie_ctx* c)
{
    choose_lanes(c);
    plan_one(c, 0, (long long)c->n_cus * 48, "PIE_K1_BLOCKS");
    plan_one(c, 1, (long long)c->n_cus * 16, "PIE_K1_BLOCKS_LIVE");
    plan_one(c, 2, (long long)c->n_cus * 32, "PIE_K1_BLOCKS_KEYED");
    plan_one(c, 3, (long long)c->n_cus * 16, "PIE_K1_BLOCKS_FINE");
}

// The staging arrays give block b of a table pass the private region [b * rows_per_block, (b + 1) * rows_per_block): a
// block never stages more records than it reads, and since the keyed pass deals rows in interleaved chunks EVERY block,
// the last one too, reads up to rows_per_block rows — so the arrays must hold blocks x rows_per_block records of the
// largest plan (up to one block's worth beyond n), not n.  (Found by the differential fuzz: a dense query on the keyed
// form overran the arrays by most of a block and trampled the neighbouring allocation.)
int ensure_sel(pie_ctx* c)
{
    long long need = c->cap_rows + 256;
    for (int k = 0; k < 4; ++k) {
        const long long span = (long long)c->plan_blocks[k] * c->plan_rows[k] + 256;
        if (span > need) need = span;
    }
    if (need <= c->sel_cap && c->slot[0].sel) return PIE_OK;
    PIE_HIP(c, hipStreamSynchronize(c->stream));
    for (Slot& s : c->slot) {
        dfree(s.sel);
        dfree(s.sel_rank);
    }
    c->sel_cap = 0;
    need += need / 16 + 65536; // room for the plans of a table that grows inside its capacity
    for (Slot& s : c->slot) {
        PIE_HIP(c, hipMalloc(&s.sel, (size_t)need * sizeof(SelRec)));
        PIE_HIP(c, hipMalloc(&s.sel_rank, (size_t)need * 4));
    }
    c->sel_cap = need;
    return PIE_OK;
}

// layout of a slot's span (all parts 128-byte aligned, total a multiple of 16 bytes so K2 can zero it as int4)
size_t span_counts_bytes(const pie_ctx* c) { return ((((size_t)c->cap_users + 32) * 4 + 127) / 128) * 128; } // +32: the transposed histogram rounds U up to a multiple of 32
size_t span_tiles_bytes(const pie_ctx* c) { return ((((size_t)c->cap_users / 256 + 2) * 8 + 127) / 128) * 128; } // sized for the smallest tile shape
size_t span_parts_bytes() { return (size_t)kPartMax * 4; }
size_t span_stats_bytes() { return (size_t)kSummaryBytes + (size_t)kStatSlots * sizeof(StatSlot); } // Summary (padded) + the K1 row-statistics slots
size_t counts_span(const pie_ctx* c) { return span_counts_bytes(c) + span_tiles_bytes(c) + span_parts_bytes() + 128 + span_stats_bytes(); }

// everything that depends on the number of users of the resident table (within the allocated capacity)
void set_user_count(pie_ctx* c, int n_users)
{
    c->n_users = n_users;
    c->n_tiles = (n_users + kScanTile - 1) / kScanTile;
    c->part_shift = -1;
    for (int sh = 0; (1 << sh) <= kPartRange; ++sh) {
        if ((((long long)n_users - 1) >> sh) + 1 <= kPartMax) { c->part_shift = sh; break; }
    }
    c->n_parts = c->part_shift >= 0 ? (int)((((long long)n_users - 1) >> c->part_shift) + 1) : 0;
}

// Make room for n rows / n_users users.  keep_rows > 0: the first keep_rows rows of the resident columns survive
// a re-allocation (append path; capacity grows geometrically so appends are amortised O(1) per row).
int ensure_capacity(pie_ctx* c, long long n, int n_users, long long keep_rows = 0)
{
    if (n < 0 || n >= (1LL << 31) - 1) return fail(c, PIE_E_INVAL, "row count %lld outside [0, 2^31 - 1)", n);
    if (n_users < 1) return fail(c, PIE_E_INVAL, "n_users must be >= 1 (got %d)", n_users);
    if (c->n_flight || c->b_flight) return fail(c, PIE_E_STATE, "table change while a scan is in flight");
    int rc = sync_all(c);
    if (rc) return rc;
    ord_invalidate(c, keep_rows == 0);
    long long rows = n > 0 ? n : 1;
    if (rows > c->cap_rows || n_users > c->cap_users) {
        int users = n_users;
        if (keep_rows > 0) { // geometric growth for the append path
            if (rows > c->cap_rows) rows = rows > 2 * c->cap_rows ? rows : 2 * c->cap_rows;
            else rows = c->cap_rows;
            if (n_users > c->cap_users) users = n_users > 2 * c->cap_users ? n_users : 2 * c->cap_users;
            else users = c->cap_users;
            if (rows >= (1LL << 31)) rows = (1LL << 31) - 1;
        }
        long long *old_s = c->d_start, *old_e = c->d_end;
        int *old_u = c->d_user, *old_d = c->d_disc;
        if (keep_rows > 0) { c->d_start = c->d_end = nullptr; c->d_user = c->d_disc = nullptr; }
        else old_s = old_e = nullptr, old_u = old_d = nullptr;
        free_table(c);
        PIE_HIP(c, hipMalloc(&c->d_start, rows * 8));
        PIE_HIP(c, hipMalloc(&c->d_end, rows * 8));
        PIE_HIP(c, hipMalloc(&c->d_user, rows * 4));
        PIE_HIP(c, hipMalloc(&c->d_disc, rows * 4));
        // derived columns, rebuilt by build_keys after every (re)allocation; a table too large to carry them (19 B/row)
        // simply runs without the keyed form
        if (hipMalloc(&c->d_key, rows * sizeof(lkey_t) + 64) != hipSuccess || hipMalloc(&c->d_pay, rows * sizeof(PayRec)) != hipSuccess ||
            hipMalloc(&c->d_fkey, rows * sizeof(fkey_t) + 64) != hipSuccess) {
            (void)hipGetLastError();
            dfree(c->d_key);
            dfree(c->d_pay);
            dfree(c->d_fkey);
        }
        if (keep_rows > 0 && old_s) {
            PIE_HIP(c, hipMemcpyAsync(c->d_start, old_s, keep_rows * 8, hipMemcpyDeviceToDevice, c->stream));
            PIE_HIP(c, hipMemcpyAsync(c->d_end, old_e, keep_rows * 8, hipMemcpyDeviceToDevice, c->stream));
            PIE_HIP(c, hipMemcpyAsync(c->d_user, old_u, keep_rows * 4, hipMemcpyDeviceToDevice, c->stream));
            PIE_HIP(c, hipMemcpyAsync(c->d_disc, old_d, keep_rows * 4, hipMemcpyDeviceToDevice, c->stream));
            PIE_HIP(c, hipStreamSynchronize(c->stream));
            (void)hipFree(old_s); (void)hipFree(old_e); (void)hipFree(old_u); (void)hipFree(old_d);
        }
        const long long max_blocks = rows / (kUnitRows * 8 * kK1Waves) + 2;
        PIE_HIP(c, hipMalloc(&c->d_blk_off, (max_blocks * kK1Waves + 8) * 8)); // per-wave prefix of the expired queue
        c->cap_users = users; // counts_span() below reads it
        for (Slot& s : c->slot) {
            PIE_HIP(c, hipMalloc(&s.offsets, ((size_t)users + 1) * 8));
            PIE_HIP(c, hipMalloc(&s.counts_ord, ((size_t)users + 1) * 4));
            // +256: K3 fetches a region's first 256 records before it knows the count
            PIE_HIP(c, hipMalloc(&s.blk_count, (max_blocks * kK1Waves + 8) * 4));
            PIE_HIP(c, hipMalloc(&s.blk_hot_base, (size_t)(max_blocks + 8) * kHotMax * 4));
            PIE_HIP(c, hipMalloc(&s.hot_list, (size_t)kHotMax * 4));
            PIE_HIP(c, hipMalloc(&s.over_list, ((size_t)users + 16) * 4));
            PIE_HIP(c, hipMalloc(&s.bkt, rows * sizeof(BktRec)));
            if ((size_t)users * kTinyMax * sizeof(BktRec) <= kDirectMaxBytes) PIE_HIP(c, hipMalloc(&s.direct, (size_t)users * kTinyMax * sizeof(BktRec)));
            c->dshift = c->dshift_want = 4;
            PIE_HIP(c, hipMalloc(&s.out_idx, rows * 4));
            PIE_HIP(c, hipMalloc(&s.seg_list, ((size_t)users + rows / kSegMax + 16) * sizeof(Segment)));
            PIE_HIP(c, hipMalloc(&s.small_list, ((size_t)users + 16) * sizeof(Segment)));
            PIE_HIP(c, hipMalloc(&s.big_list, ((size_t)users + 16) * 4));
            PIE_HIP(c, hipMalloc(&s.part_rec, (size_t)kPartMax * kPartCap * sizeof(SelRec)));
        }
        for (char*& sp : c->span) PIE_HIP(c, hipMalloc(&sp, counts_span(c)));
        c->cap_rows = rows;
    }
    c->n = n;
    c->n_users = n_users;
    c->n_tiles = (n_users + kScanTile - 1) / kScanTile;
    if (keep_rows == 0) {
        // a new table: nothing is known about it.  An append keeps what the last scans observed (live fraction, skew,
        // clustering): a few new rows do not change the picture, and one scan corrects it if they do
        c->live_frac = -1;
        c->batch_poor = false;
        c->ord_lists_only = false;
        if (!c->run_shift_pinned) c->run_shift = 0;
        c->hot_bucket = false;
        c->clustered = false;
        c->hot.n = 0;
        c->hot_seen = 0;
        c->hot_age = 0;
        c->last_m = -1;
    }
    c->fast_enabled = c->fast_env;
    set_user_count(c, n_users);
    c->res = nullptr;
    for (Slot& s : c->slot) s.have_result = false;
    // all three spans start clean; from here on every K2 zeroes the span of the scan after it
    for (char* sp : c->span) PIE_HIP(c, hipMemsetAsync(sp, 0, counts_span(c), c->stream));
    c->span_next = 0;
    plan_k1(c);
    return ensure_sel(c);
}

// pinned host + device staging blocks of at least `bytes` (one pair per context; the context is used by one thread)
int ensure_stage(pie_ctx* c, size_t bytes)
{
    if (bytes <= c->stage_bytes) return PIE_OK;
    size_t want = c->stage_bytes ? c->stage_bytes : (size_t)64 << 10;
    while (want < bytes) want *= 2;
    PIE_HIP(c, hipStreamSynchronize(c->stream));
    if (c->h_stage) (void)hipHostFree(c->h_stage);
    dfree(c->d_stage);
    c->h_stage = nullptr;
    c->stage_bytes = 0;
    PIE_HIP(c, hipHostMalloc(&c->h_stage, want, hipHostMallocDefault));
    PIE_HIP(c, hipMalloc(&c->d_stage, want));
    c->stage_bytes = want;
    return PIE_OK;
}

// one of the two staging areas of the queued mutations, free to be written: *out
int async_stage(pie_ctx* c, size_t bytes, pie_ctx::AsyncStage** out)
{
    pie_ctx::AsyncStage& a = c->astage[c->astage_next];
    c->astage_next ^= 1;
    if (a.pending) { // the copy out of it was queued two mutations ago
        PIE_HIP(c, hipEventSynchronize(a.ev));
        a.pending = false;
    }
    if (!a.ev) PIE_HIP(c, hipEventCreateWithFlags(&a.ev, hipEventDisableTiming));
    if (bytes > a.bytes) {
        size_t want = a.bytes ? a.bytes : (size_t)64 << 10;
        while (want < bytes) want *= 2;
        if (a.h) (void)hipHostFree(a.h);
        if (a.d) { PIE_HIP(c, hipStreamSynchronize(c->stream)); (void)hipFree(a.d); } // a kernel queued earlier may still read it
        a.h = nullptr; a.d = nullptr; a.bytes = 0;
        PIE_HIP(c, hipHostMalloc(&a.h, want, hipHostMallocDefault));
        PIE_HIP(c, hipMalloc(&a.d, want));
        a.bytes = want;
    }
    *out = &a;
    return PIE_OK;
}

int validate_users(pie_ctx* c, long long row0 = 0)
{
    if (c->n - row0 <= 0) return PIE_OK;
    PIE_HIP(c, hipMemsetAsync(c->d_summary, 0, sizeof(Summary), c->stream));
    const int grid = c->n_cus * 8;
    hipLaunchKernelGGL(k_validate_users, dim3(grid), dim3(256), 0, c->stream, c->d_user + row0, c->n - row0, c->n_users,
                       &c->d_summary->bad_rows);
    PIE_HIP(c, hipGetLastError());
    PIE_HIP(c, hipMemcpyAsync(c->h_summary, c->d_summary, sizeof(Summary), hipMemcpyDeviceToHost, c->stream));
    PIE_HIP(c, hipStreamSynchronize(c->stream));
    if (c->h_summary->bad_rows)
        return fail(c, PIE_E_INVAL, "%u rows carry a user id outside [0, %d)", c->h_summary->bad_rows, c->n_users);
    return PIE_OK;
}

// Liveness-key column for rows [row0, n).  row0 == 0, a re-allocated table, or `rebuild` re-derive (base, shift) from the
// range of the live `end` values and key every row; otherwise the appended rows are keyed under the current parameters
// (out-of-range values clamp; the column stays exact, only less selective).
int build_keys(pie_ctx* c, long long row0, bool rebuild = false)
{
    if (!c->d_key || !c->d_pay || !c->d_fkey) { c->key_ok = false; return PIE_OK; }
    const bool write_pay = !(rebuild && c->key_ok); // a refit of the key leaves the (immutable) payload alone
    if (c->n == 0) { c->key_ok = true; c->key_base = 0; c->key_shift = 0; c->key_dirty = false; return PIE_OK; }
    hipStream_t s = c->stream;
    const int grid = c->n_cus * 8;
    const bool full_build = row0 == 0 || !c->key_ok || rebuild;
    // the ordered run carries copies of both keys: a refit of the table's keys refits them from its copy of `end` (below);
    // a first build means a new table, whose run went with the old one
    if (full_build && !(rebuild && c->key_ok && c->ord.valid && c->ord.rows == c->n)) ord_invalidate(c);
    timespec tb0{};
    if (full_build) { // index_build_ms of pie_table_info: everything from here to the last key kernel's completion
        PIE_HIP(c, hipStreamSynchronize(s));
        clock_gettime(CLOCK_MONOTONIC, &tb0);
    }
    if (row0 == 0 || !c->key_ok || rebuild) {
        const long long init[2] = {INT64_MAX, INT64_MIN};
        long long got[2];
        PIE_HIP(c, hipMemcpyAsync(c->d_range, init, sizeof init, hipMemcpyHostToDevice, s));
        hipLaunchKernelGGL(k_end_range, dim3(grid), dim3(256), 0, s, c->d_end, c->n, c->d_range);
        PIE_HIP(c, hipGetLastError());
        PIE_HIP(c, hipMemcpyAsync(got, c->d_range, sizeof got, hipMemcpyDeviceToHost, s));
        PIE_HIP(c, hipStreamSynchronize(s));
        c->key_base = 0;
        c->key_shift = 0;
        if (got[0] <= got[1]) {
            c->key_base = got[0];
            const unsigned long long span = (unsigned long long)got[1] - (unsigned long long)got[0];
            while (c->key_shift < 63 && (span >> c->key_shift) >= (unsigned long long)(kKeyMax - 1u)) c->key_shift++;
        }
        row0 = 0;
        c->key_dirty = false;
        c->key_poor = false;
        c->key_rebuild = false;
    } else {
        c->key_dirty = true;
    }
    hipLaunchKernelGGL(k_build_key, dim3(grid), dim3(256), 0, s, c->d_end, row0, c->n, c->key_base, c->key_shift, c->d_key,
                       c->d_start, c->d_user, c->d_disc, write_pay ? c->d_pay : (PayRec*)nullptr);
    PIE_HIP(c, hipGetLastError());
    if (row0 == 0) {
        // fine key: base = lower edge of the histogram bin that holds the 90th percentile of the 15-bit keys (tombstones
        // and rows below the base count as key 0), shift = smallest that keeps the largest `end` under the clamp
        std::vector<unsigned int> hist(kKeyHistBins);
        PIE_HIP(c, hipMemsetAsync(c->d_hist, 0, kKeyHistBins * sizeof(unsigned int), s));
        hipLaunchKernelGGL(k_key_hist, dim3(c->n_cus), dim3(1024), 0, s, c->d_key, c->n, c->d_hist);
        PIE_HIP(c, hipGetLastError());
        PIE_HIP(c, hipMemcpyAsync(hist.data(), c->d_hist, kKeyHistBins * sizeof(unsigned int), hipMemcpyDeviceToHost, s));
        PIE_HIP(c, hipStreamSynchronize(s));
        c->key_hist = hist;
        c->key_hist_rows = c->n;
        c->key_hist_above.assign((size_t)kKeyHistBins + 1, 0ull);
        for (int b = kKeyHistBins - 1; b >= 0; --b) c->key_hist_above[(size_t)b] = c->key_hist_above[(size_t)b + 1] + hist[(size_t)b];
        unsigned long long cum = 0;
        int bin = 0, top_bin = 0;
        for (int b = 0; b < kKeyHistBins; ++b)
            if (hist[b]) top_bin = b;
        const unsigned long long want = (unsigned long long)((double)c->n * 0.9);
        for (bin = 0; bin < kKeyHistBins - 1; ++bin) {
            if (cum + hist[bin] > want) break;
            cum += hist[bin];
        }
        if (bin > top_bin) bin = top_bin;
        // key k >= 1 covers [base + ((k-1) << shift), base + (k << shift)); bin b holds keys 8b .. 8b+7
        c->fkey_base = bin == 0 ? c->key_base : (long long)((unsigned long long)c->key_base + (((unsigned long long)bin * 8ull - 1ull) << c->key_shift));
        const unsigned long long top_edge = (unsigned long long)c->key_base + (((unsigned long long)top_bin * 8ull + 8ull) << c->key_shift);
        const unsigned long long fspan = top_edge - (unsigned long long)c->fkey_base;
        c->fkey_shift = 0;
        while (c->fkey_shift < 63 && (fspan >> c->fkey_shift) >= (unsigned long long)(kFineKeyMax - 1u)) c->fkey_shift++;
        c->fkey_poor = false;
    }
    hipLaunchKernelGGL(k_build_fine_key, dim3(grid), dim3(256), 0, s, c->d_end, row0, c->n, c->fkey_base, c->fkey_shift, c->d_fkey);
    PIE_HIP(c, hipGetLastError());
    if (rebuild && c->ord.valid) { // filler positions hold end = 0 (or below every base): key 0 under any parameters
        hipLaunchKernelGGL(k_build_key, dim3(grid), dim3(256), 0, s, c->ord.end, 0LL, c->ord.n, c->key_base, c->key_shift, c->ord.key,
                           (const long long*)nullptr, (const int*)nullptr, (const int*)nullptr, (PayRec*)nullptr);
        hipLaunchKernelGGL(k_build_fine_key, dim3(grid), dim3(256), 0, s, c->ord.end, 0LL, c->ord.n, c->fkey_base, c->fkey_shift, c->ord.fkey);
        PIE_HIP(c, hipGetLastError());
    }
    c->key_ok = true;
    if (full_build) {
        PIE_HIP(c, hipStreamSynchronize(s));
        timespec tb1{};
        clock_gettime(CLOCK_MONOTONIC, &tb1);
        c->index_build_ms = (double)(tb1.tv_sec - tb0.tv_sec) * 1e3 + (double)(tb1.tv_nsec - tb0.tv_nsec) * 1e-6;
    }
    return PIE_OK;
}

unsigned host_key_of(const pie_ctx* c, long long e)
{
    if (e < c->key_base) return 0u;
    const unsigned long long k = ((unsigned long long)e - (unsigned long long)c->key_base) >> c->key_shift;
    return k >= (unsigned long long)(kKeyMax - 1u) ? kKeyMax : (unsigned)k + 1u;
}

// rows whose 15-bit key lies in the histogram bin of key(now) or above: an upper bound on the rows live at `now`
// (valid while the histogram describes the table: !key_dirty && key_hist_rows == n)
unsigned long long rows_keyed_at_or_above(const pie_ctx* c, long long now)
{
    const size_t bin = (size_t)(host_key_of(c, now) >> 3);
    return bin < c->key_hist_above.size() ? c->key_hist_above[bin] : 0ull;
}

unsigned host_fine_key_of(const pie_ctx* c, long long e)
{
    if (e < c->fkey_base) return 0u;
    const unsigned long long k = ((unsigned long long)e - (unsigned long long)c->fkey_base) >> c->fkey_shift;
    return k >= (unsigned long long)(kFineKeyMax - 1u) ? kFineKeyMax : (unsigned)k + 1u;
}

int resolve_events(pie_ctx* c)
{
    if (c->ring_used == 0) return PIE_OK;
    if (c->n_flight) return PIE_OK; // events of scans in flight are not closed yet; resolved on a later call
    int rc = sync_all(c);
    if (rc) return rc;
    for (int i = 0; i < c->ring_used; ++i) {
        float k1 = 0, all = 0;
        PIE_HIP(c, hipEventElapsedTime(&k1, c->ring[i].e0, c->ring[i].e1));
        PIE_HIP(c, hipEventElapsedTime(&all, c->ring[i].e0, c->ring[i].e2));
        c->k1_ms_sum += k1;
        c->scan_ms_sum += all;
        c->n_profiled++;
    }
    c->ring_used = 0;
    return PIE_OK;
}

// K1 forms.  Variant code: bit0 nontemporal loads, bit1 late user materialisation, bit2 liveness-first form
// (streams only `end`, gathers the other columns for live rows); bits 4,5,7 = unroll (0 -> 4, 0x20 -> 2, 0x80 -> 8);
// bit6 (0x40, liveness-first only) = wave-aggregated histogram atomics for skewed users.
// PIE_K1_VARIANT pins one form (tuning / A-B runs); otherwise the form follows the live fraction that the
// previous scan on this table observed: liveness-first below kLiveFirstBelow, the streaming form above.
constexpr double kLiveFirstBelow = 0.10; // measured crossover ~0.16 live (profiles/r01_c_live_fraction_crossover.txt)

void launch_k1(pie_ctx* c, Slot& sl, hipStream_t s, long long now, long long cutoff, unsigned long long mask)
{
#define PIE_K1(UN, NT, LU)                                                                                          \
    hipLaunchKernelGGL((k_scan_compact<UN, NT, LU>), dim3(sl.k1_blocks), dim3(kK1Threads), 0, s, c->d_start, c->d_end, \
                       c->d_user, c->d_disc, c->n, sl.rows_per_block, now, cutoff, mask, c->n_users, sl.counts, sl.sel, \
                       sl.sel_rank, sl.blk_count, sl.sum, direct_of(c, sl))
#define PIE_K1L(UN, NT)                                                                                             \
    if (sl.variant & 0x40)                                                                                          \
        hipLaunchKernelGGL((k_scan_live_first<UN, NT, true>), dim3(sl.k1_blocks), dim3(kK1Threads), 0, s, c->d_start, c->d_end, \
                           c->d_user, c->d_disc, c->n, sl.rows_per_block, now, cutoff, mask, c->n_users, sl.counts, sl.sel, \
                           sl.sel_rank, sl.blk_count, sl.sum, direct_of(c, sl), sl.hot, sl.blk_hot_base);          \
    else                                                                                                            \
        hipLaunchKernelGGL((k_scan_live_first<UN, NT, false>), dim3(sl.k1_blocks), dim3(kK1Threads), 0, s, c->d_start, c->d_end, \
                       c->d_user, c->d_disc, c->n, sl.rows_per_block, now, cutoff, mask, c->n_users, sl.counts, sl.sel, \
                       sl.sel_rank, sl.blk_count, sl.sum, direct_of(c, sl), sl.hot, sl.blk_hot_base)
    if (sl.variant & 0x400) { // keyed liveness-first form
#define PIE_K1K3(UN, AG, NT, PP)                                                                                     \
        do {                                                                                                        \
            if (sl.variant & 0x800)                                                                                 \
                hipLaunchKernelGGL((k_scan_keyed<UN, AG, NT, fkey_t, PP>), dim3(sl.k1_blocks), dim3(kK1Threads), 0, s, c->d_pay, \
                                   c->d_end, c->d_fkey, c->n, sl.rows_per_block, now, host_fine_key_of(c, now), cutoff, mask, \
                                   c->n_users, sl.counts, sl.sel, sl.sel_rank, sl.blk_count, sl.sum, direct_of(c, sl), \
                                   sl.hot, sl.blk_hot_base, c->run_shift);                                          \
            else                                                                                                    \
                hipLaunchKernelGGL((k_scan_keyed<UN, AG, NT, lkey_t, PP>), dim3(sl.k1_blocks), dim3(kK1Threads), 0, s, c->d_pay, \
                                   c->d_end, c->d_key, c->n, sl.rows_per_block, now, host_key_of(c, now), cutoff, mask,       \
                                   c->n_users, sl.counts, sl.sel, sl.sel_rank, sl.blk_count, sl.sum, direct_of(c, sl), \
                                   sl.hot, sl.blk_hot_base, c->run_shift);                                          \
        } while (0)
#define PIE_K1K(UN, AG, NT)                                                                                          \
        do {                                                                                                        \
            if (sl.variant & 0x10) PIE_K1K3(UN, AG, NT, true);                                                       \
            else PIE_K1K3(UN, AG, NT, false);                                                                        \
        } while (0)
#define PIE_K1K2(UN)                                                                                                 \
        do {                                                                                                        \
            if (sl.variant & 0x40) { if (sl.variant & 1) PIE_K1K(UN, true, true); else PIE_K1K(UN, true, false); }   \
            else { if (sl.variant & 1) PIE_K1K(UN, false, true); else PIE_K1K(UN, false, false); }                   \
        } while (0)
        switch (sl.variant & 0xA0) {
        case 0x20: PIE_K1K2(2); break;
        case 0x80: PIE_K1K2(8); break;
        default: PIE_K1K2(4); break;
        }
#undef PIE_K1K2
#undef PIE_K1K3
#undef PIE_K1K
        return;
    }
    if (sl.variant == 0x101) { // group-qualified form, used only by pie_archive_queue
        hipLaunchKernelGGL((k_scan_compact<4, true, false, true>), dim3(sl.k1_blocks), dim3(kK1Threads), 0, s, c->d_start,
                           c->d_end, c->d_user, c->d_disc, c->n, sl.rows_per_block, now, cutoff, mask, c->n_users, sl.counts,
                           sl.sel, sl.sel_rank, sl.blk_count, sl.sum, direct_of(c, sl), c->d_qual);
        return;
    }
    if (sl.variant == 0x43) { // streaming form with wave-aggregated histogram atomics (rows clustered by user)
        hipLaunchKernelGGL((k_scan_compact<4, true, true, false, true>), dim3(sl.k1_blocks), dim3(kK1Threads), 0, s, c->d_start,
                           c->d_end, c->d_user, c->d_disc, c->n, sl.rows_per_block, now, cutoff, mask, c->n_users, sl.counts,
                           sl.sel, sl.sel_rank, sl.blk_count, sl.sum, direct_of(c, sl));
        return;
    }
    switch (sl.variant & ~0x40) {
    case 0x00: PIE_K1(4, false, false); break;
    case 0x01: PIE_K1(4, true, false); break;
    case 0x02: PIE_K1(4, false, true); break;
    case 0x03: PIE_K1(4, true, true); break;
    case 0x04: PIE_K1L(4, false); break;
    case 0x05: PIE_K1L(4, true); break;
    case 0x24: PIE_K1L(2, false); break;
    case 0x25: PIE_K1L(2, true); break;
    case 0x84: PIE_K1L(8, false); break;
    case 0x85: PIE_K1L(8, true); break;
    case 0x20: PIE_K1(2, false, false); break;
    case 0x21: PIE_K1(2, true, false); break;
    case 0x22: PIE_K1(2, false, true); break;
    case 0x23: PIE_K1(2, true, true); break;
    case 0x80: PIE_K1(8, false, false); break;
    case 0x81: PIE_K1(8, true, false); break;
    case 0x82: PIE_K1(8, false, true); break;
    case 0x83: PIE_K1(8, true, true); break;
    default: sl.variant = 0x03; PIE_K1(4, true, true); break;
    }
#undef PIE_K1
#undef PIE_K1L
}

// per-bucket order of the buckets of <= kTinyMax rows; with direct slots it depends on K2 only (not on K3, not on the
// host), so it is queued right behind K2
void launch_sort_tiny(pie_ctx* c, Slot& sl, hipStream_t s)
{
    const int tiny_blocks = (c->n_users + 255) / 256;
    hipLaunchKernelGGL(k_sort_tiny, dim3(tiny_blocks), dim3(256), 0, s, sl.counts_ord, sl.offsets, c->n_users, sl.bkt, direct_of(c, sl), sl.out_idx, sl.hot);
}

// K2 (+ the order of the tiny buckets).  Three shapes: fused (one user per thread; offsets and tiny-bucket order in one
// kernel) when buckets have direct slots and the tile count suits the all-predecessors look-back; otherwise the
// 2048-user tiles, followed by the tiny-bucket kernel right away (direct slots) or after K3 (staged route).
constexpr int kOrderMaxTiles = 2048; // fused form: few enough tiles for the all-predecessors look-back

// a bucket is "hot" when it holds more than 1/256 of the rows the previous scan selected (and enough of them that
// same-address histogram atomics hurt): such users are reported for the next scan's hot set
int hot_threshold(const pie_ctx* c)
{
    return (c->last_m >= 65536 && !c->d_qual) ? (int)std::min<long long>(c->last_m / 256, 0x7FFFFFFF) : 0;
}

void launch_k2(pie_ctx* c, Slot& sl, hipStream_t s, int4* zero_span, long long zero_vec16)
{
    if (!zero_span) zero_vec16 = 0; // a scan outside the span rotation (the ordered run's build) zeroes nothing
    const int hot_thr = hot_threshold(c);
    const int ob = c->order_block;
    const int order_tiles = (c->n_users + ob - 1) / ob;
    if (sl.direct && order_tiles <= kOrderMaxTiles && !c->no_fused_order) {
#define PIE_K2O(B)                                                                                                          \
    hipLaunchKernelGGL((k_offsets<1, true, B>), dim3(order_tiles), dim3(B), 0, s, sl.counts, sl.counts_ord, c->n_users, sl.tile_pub, sl.ctl,   \
                       sl.offsets, sl.seg_list, sl.small_list, sl.big_list, sl.sum, sl.h_sum_dev, sl.seq, zero_span, zero_vec16, \
                       direct_of(c, sl), sl.bkt, sl.out_idx, sl.msg, sl.msg_u_pad, sl.msg_cap, sl.msg_counts, sl.hot, hot_thr, sl.hot_list, sl.over_list)
        if (ob == 256) PIE_K2O(256);
        else if (ob == 512) PIE_K2O(512);
        else PIE_K2O(1024);
#undef PIE_K2O
        sl.msg_by_k2 = sl.msg != nullptr;
        return;
    }
    hipLaunchKernelGGL((k_offsets<8, false, 256>), dim3(c->n_tiles), dim3(256), 0, s, sl.counts, sl.counts_ord, c->n_users, sl.tile_pub, sl.ctl,
                       sl.offsets, sl.seg_list, sl.small_list, sl.big_list, sl.sum, sl.h_sum_dev, sl.seq, zero_span, zero_vec16,
                       direct_of(c, sl), sl.bkt, (int*)nullptr, (int*)nullptr, 0, 0LL, (int*)nullptr, sl.hot, hot_thr, sl.hot_list, sl.over_list);
    sl.msg_by_k2 = false;
    if (sl.direct) launch_sort_tiny(c, sl, s);
}

// K2 of `tail` can ride in another scan's launch: fused form (direct slots), few enough 256-user tiles.  (Beside a
// table pass K2 takes about twice as long, so a caller that wants a scan's summary early should not begin the next scan
// first; the exchange driver in shard.py does its host work in that window instead.)
bool tail_can_ride(const pie_ctx* c, const Slot& tail)
{
    const int tiles = (c->n_users + kK1Threads - 1) / kK1Threads;
    return tail.direct != nullptr && tiles <= kOrderMaxTiles && !c->no_fused_order && !c->no_ride && !tail.fast && !c->d_qual;
}

// the default keyed forms (unroll 8, nontemporal loads, with or without wave aggregation; no pipelining) exist with a tail
bool keyed_can_carry(const pie_ctx* c, const Slot& sl)
{
    (void)c;
    return (sl.variant & ~0x840) == 0x485;
}

// ... and so do the plain streaming forms (nontemporal loads, unroll 4, with or without late user materialisation): the
// every-byte scan and dense queries on tables without an ordered run.  PIE_STREAM_RIDE=0 turns it off (A/B runs).
bool stream_can_carry(const pie_ctx* c, const Slot& sl)
{
    if (c->d_qual || !(sl.variant == 0x01 || sl.variant == 0x03)) return false;
    const char* v = getenv("PIE_STREAM_RIDE"); // read per scan (only on this rare form): in-process A/B runs flip it
    return !(v && atoi(v) == 0);
}

void launch_keyed_with_tail(pie_ctx* c, Slot& sl, Slot& tail, hipStream_t s, long long now, long long cutoff, unsigned long long mask)
{
    OffsetsArgs t;
    t.counts = tail.counts; t.counts_ord = tail.counts_ord; t.n_users = c->n_users; t.tile_pub = tail.tile_pub; t.ctl = tail.ctl;
    t.offsets = tail.offsets; t.seg_list = tail.seg_list; t.small_list = tail.small_list; t.big_list = tail.big_list;
    t.summary = tail.sum; t.host = tail.h_sum_dev; t.seq = tail.seq; t.zero_span = tail.zero_span;
    t.zero_vec16 = (long long)(counts_span(c) / 16); t.direct = direct_of(c, tail); t.bkt = tail.bkt; t.out_idx = tail.out_idx;
    t.msg = tail.msg; t.u_pad = tail.msg_u_pad; t.msg_cap = tail.msg_cap; t.msg_counts = tail.msg_counts; t.hot = tail.hot; t.hot_thr = hot_threshold(c);
    t.hot_list = tail.hot_list; t.over_list = tail.over_list;
    t.n_tail = (c->n_users + kK1Threads - 1) / kK1Threads;
    tail.msg_by_k2 = tail.msg != nullptr;
    const unsigned grid = (unsigned)(sl.k1_blocks + t.n_tail);
    if (!(sl.variant & 0x400)) { // a streaming form carries the tail
        StreamArgs a;
        a.start = c->d_start; a.end = c->d_end; a.user = c->d_user; a.disc = c->d_disc; a.n = c->n; a.rows_per_block = sl.rows_per_block;
        a.now = now; a.cutoff = cutoff; a.mask = mask; a.n_users = c->n_users; a.counts = sl.counts; a.sel = sl.sel; a.sel_rank = sl.sel_rank;
        a.blk_count = sl.blk_count; a.summary = sl.sum; a.direct = direct_of(c, sl);
        if (sl.variant & 2) hipLaunchKernelGGL((k_scan_compact_with_tail<true>), dim3(grid), dim3(kK1Threads), 0, s, a, t);
        else hipLaunchKernelGGL((k_scan_compact_with_tail<false>), dim3(grid), dim3(kK1Threads), 0, s, a, t);
        return;
    }
#define PIE_RIDE(KT, KEYPTR, NOWKEY)                                                                                     \
    do {                                                                                                                \
        KeyedArgs<KT> a;                                                                                                \
        a.pay = c->d_pay; a.end = c->d_end; a.key = KEYPTR; a.n = c->n; a.rows_per_block = sl.rows_per_block; a.now = now; \
        a.now_key = NOWKEY; a.cutoff = cutoff; a.mask = mask; a.n_users = c->n_users; a.counts = sl.counts; a.sel = sl.sel;  \
        a.sel_rank = sl.sel_rank; a.blk_count = sl.blk_count; a.summary = sl.sum; a.direct = direct_of(c, sl); a.hot = sl.hot; \
        a.blk_hot_base = sl.blk_hot_base; a.run_shift = c->run_shift;                                                   \
        if (sl.variant & 0x40) hipLaunchKernelGGL((k_scan_keyed_with_tail<8, true, KT, true>), dim3(grid), dim3(kK1Threads), 0, s, a, t);  \
        else hipLaunchKernelGGL((k_scan_keyed_with_tail<8, true, KT, false>), dim3(grid), dim3(kK1Threads), 0, s, a, t);   \
    } while (0)
    if (sl.variant & 0x800) PIE_RIDE(fkey_t, c->d_fkey, host_fine_key_of(c, now));
    else PIE_RIDE(lkey_t, c->d_key, host_key_of(c, now));
#undef PIE_RIDE
}

// Head of a scan: K1 and K2 on the stream (plus the tiny-bucket order kernel when buckets have direct slots).  No host wait.
// ------------------------------------------------------------------------------------------------ the ordered run (pie_ordered.h)

int scan_begin(pie_ctx* c, long long now, long long cutoff, int* msg, int msg_u_pad, long long msg_cap, int* msg_counts);
int scan_finish(pie_ctx* c, Slot* which = nullptr);

size_t ord_sum_bytes() { return span_stats_bytes() + 128; } // Summary + row-statistics slots, then OrdCtl

// arrays for the table's current capacity; no room is not an error (the general path serves every query)
int ord_alloc(pie_ctx* c)
{
    OrderedRun& o = c->ord;
    if (o.pay && o.cap >= c->cap_rows && o.cap_users >= c->cap_users) return PIE_OK;
    ord_free(c);
    // positions: every row of capacity, a sixteenth more and four per user as spare slots; then whole chunks of zero keys
    const size_t positions = (size_t)c->cap_rows + (size_t)c->cap_rows / 16 + (size_t)ord_spare(c) * (size_t)c->cap_users + 64;
    const size_t padded = ((positions + 1023) / 1024) * 1024 + 1024;
    if (padded + kOrdTile >= ((size_t)1 << 31)) { // positions are 31-bit (ring entries, staging records, pos[])
        o.no_room = true;
        return PIE_OK;
    }
    // a scan stages one 4-byte entry per position in the slot's record staging (sel): a table with far more users than rows
    // (four spare slots each) does not fit there, and has no use for the run anyway
    if ((padded + kOrdTile) * 4 > (size_t)c->sel_cap * sizeof(SelRec)) {
        o.no_room = true;
        return PIE_OK;
    }
    const size_t units = ((padded / 512 + 8 + 1023) / 1024) * 1024;              // the smallest unit is a 512-position chunk; whole groups
    const size_t tiles = padded / kOrdTile + 2;
    const bool ok = hipMalloc(&o.pay, padded * sizeof(OrdRec)) == hipSuccess && hipMalloc(&o.end, padded * 8) == hipSuccess &&
                    hipMalloc(&o.key, padded * sizeof(lkey_t)) == hipSuccess && hipMalloc(&o.fkey, padded * sizeof(fkey_t)) == hipSuccess &&
                    hipMalloc(&o.pos, (size_t)c->cap_rows * 4 + 64) == hipSuccess &&
                    hipMalloc(&o.uoff, ((size_t)c->cap_users + 1) * 8) == hipSuccess &&
                    hipMalloc(&o.ufill, ((size_t)c->cap_users + 1) * 4) == hipSuccess &&
                    hipMalloc(&o.pend, ((size_t)c->cap_users + 1) * 4) == hipSuccess && hipMalloc(&o.placed, 2 * (size_t)kOrdAppendMax * 4) == hipSuccess &&
                    hipMalloc(&o.bhead, ((size_t)c->cap_users + 1) * 4) == hipSuccess && hipMalloc(&o.bnext, (size_t)kOrdAppendMax * 4) == hipSuccess &&
                    hipMalloc(&o.unit_count[0], units * 4) == hipSuccess && hipMalloc(&o.unit_count[1], units * 4) == hipSuccess &&
                    hipMalloc(&o.unit_local, units * 4) == hipSuccess && hipMalloc(&o.group_sum, (units / 1024 + 2) * 8) == hipSuccess &&
                    hipMalloc(&o.group_base, (units / 1024 + 2) * 8) == hipSuccess &&
                    hipMalloc(&o.tile_ballot, tiles * kOrdSlices * 8) == hipSuccess &&
                    hipMalloc(&o.tile_prefix, tiles * kOrdSlices * 4) == hipSuccess &&
                    hipMalloc(&o.sum[0], ord_sum_bytes()) == hipSuccess && hipMalloc(&o.sum[1], ord_sum_bytes()) == hipSuccess;
    if (!ok) {
        (void)hipGetLastError();
        ord_free(c);
        o.no_room = true;
        return PIE_OK;
    }
    o.cap = c->cap_rows;
    o.pos_cap = (long long)positions;
    o.cap_users = c->cap_users;
    o.units_cap = (long long)units;
    return PIE_OK;
}

// Build the run: ONE all-selecting scan on the general path (every row that any query can ever select: end above
// INT64_MIN, discipline inside the table) gives the rows in (user, start, row) order and the per-user segment starts;
// a gather makes the run's columns from them.  Nothing the scan learns about the table is kept: it is not a query.
int build_ordered(pie_ctx* c)
{
    OrderedRun& o = c->ord;
    if (c->n_flight > 1) return fail(c, PIE_E_STATE, "ordered run: two scans are in flight");
    hipStream_t s = c->stream;
    PIE_HIP(c, hipStreamSynchronize(s));
    timespec t0{};
    clock_gettime(CLOCK_MONOTONIC, &t0);
    int rc = ord_alloc(c);
    if (rc || !o.pay) return rc;
    struct Keep {
        unsigned long long mask; int n_disc, variant, dshift_want, next_slot, mode, run_shift, bdshift_want;
        bool pinned, fast, hot_bucket, clustered, key_poor, fkey_poor, profiling, last_was_batch, batch_poor;
        double live_frac; long long last_m; HotSet hot; unsigned hot_seen, hot_age; Slot* res;
    } k{c->disc_mask, c->n_disc, c->k1_variant, c->dshift_want, c->next_slot, o.mode, c->run_shift, c->bdshift_want,
        c->k1_pinned, c->fast_enabled, c->hot_bucket, c->clustered, c->key_poor, c->fkey_poor, c->profiling, c->last_was_batch, c->batch_poor,
        c->live_frac, c->last_m, c->hot, c->hot_seen, c->hot_age, c->res};
    c->disc_mask = ~0ull;
    c->n_disc = 64;
    c->k1_pinned = true;
    c->k1_variant = 0x43; // streaming, histogram atomics aggregated per wave: bounded whatever the row order (a user-clustered
                          // table would otherwise send 64 lanes to one counter, 10^8 times)
    c->fast_enabled = false;
    c->profiling = false;
    o.mode = 0;           // the scan below must not come back here
    // The build's scan stands OUTSIDE the span rotation.  A caller's scan may be in flight: its tail (launched at its finish,
    // i.e. after everything queued here) still reads its span, which a regular scan's K2 would zero for the scan after the
    // next.  So this scan zeroes nothing, its own span is wiped when it is done, and the rotation continues where it was.
    const int keep_span_next = c->span_next;
    c->ord_building = true;
    // one scan of the caller's may be in flight (a pipelined caller always has one): the build takes the free slot, and
    // finishes THAT slot; the caller's scan stays the oldest in flight
    Slot* built = &c->slot[c->next_slot];
    rc = scan_begin(c, INT64_MIN, INT64_MIN, nullptr, 0, 0, nullptr);
    if (rc == PIE_OK) rc = scan_finish(c, built);
    c->ord_building = false;
    (void)hipMemsetAsync(c->span[keep_span_next], 0, counts_span(c), s);
    c->span_next = keep_span_next;
    c->disc_mask = k.mask; c->n_disc = k.n_disc; c->k1_pinned = k.pinned; c->k1_variant = k.variant; c->fast_enabled = k.fast;
    c->profiling = k.profiling; o.mode = k.mode; c->dshift_want = k.dshift_want; c->hot_bucket = k.hot_bucket; c->clustered = k.clustered;
    c->key_poor = k.key_poor; c->fkey_poor = k.fkey_poor; c->live_frac = k.live_frac; c->last_m = k.last_m; c->hot = k.hot;
    c->hot_seen = k.hot_seen; c->hot_age = k.hot_age; c->last_was_batch = k.last_was_batch; c->run_shift = k.run_shift;
    c->bdshift_want = k.bdshift_want; c->batch_poor = k.batch_poor;
    if (rc) { // the caller's last result and slot order are as they were; the failed scan left nothing in flight of its own
        built->have_result = false;
        c->res = (k.res && k.res != built) ? k.res : nullptr;
        c->next_slot = k.next_slot;
        return rc;
    }
    const long long m = (long long)built->last.m;
    // segments: user u's rows, then spare slots for the rows to come (k_ord_append); users that do not exist yet get four
    const int seg_users = c->cap_users;
    std::vector<int> cnt((size_t)seg_users + 1, 0);
    PIE_HIP(c, hipMemcpyAsync(cnt.data(), built->counts_ord, (size_t)c->n_users * 4, hipMemcpyDeviceToHost, s));
    PIE_HIP(c, hipStreamSynchronize(s));
    std::vector<long long> seg((size_t)seg_users + 1);
    long long total = 0;
    const int spare = ord_spare(c);
    for (int u = 0; u < seg_users; ++u) {
        seg[(size_t)u] = total;
        total += (long long)cnt[(size_t)u] + cnt[(size_t)u] / 16 + spare;
    }
    seg[(size_t)seg_users] = total;
    if (total > o.pos_cap) { o.no_room = true; return PIE_OK; } // cannot happen by ord_alloc's sizing; stay on the general path
    const size_t padded = (((size_t)o.pos_cap + 1023) / 1024) * 1024 + 1024;
    PIE_HIP(c, hipMemcpyAsync(o.uoff, seg.data(), ((size_t)seg_users + 1) * 8, hipMemcpyHostToDevice, s));
    PIE_HIP(c, hipMemcpyAsync(o.ufill, cnt.data(), ((size_t)seg_users + 1) * 4, hipMemcpyHostToDevice, s));
    PIE_HIP(c, hipMemsetAsync(o.pend, 0, ((size_t)seg_users + 1) * 4, s));
    PIE_HIP(c, hipMemsetAsync(o.bhead, 0xFF, ((size_t)seg_users + 1) * 4, s));
    PIE_HIP(c, hipMemsetAsync(o.pos, 0xFF, (size_t)o.cap * 4, s));
    PIE_HIP(c, hipMemsetAsync(o.pay, 0xFF, padded * sizeof(OrdRec), s)); // filler: discipline -1, never selected
    hipLaunchKernelGGL(k_fill_ll, dim3(c->n_cus * 8), dim3(256), 0, s, o.end, (long long)padded, (long long)INT64_MIN);
    PIE_HIP(c, hipMemsetAsync(o.key, 0, padded * sizeof(lkey_t), s));
    PIE_HIP(c, hipMemsetAsync(o.fkey, 0, padded * sizeof(fkey_t), s));
    PIE_HIP(c, hipMemsetAsync(o.unit_count[0], 0, (size_t)o.units_cap * 4, s));
    PIE_HIP(c, hipMemsetAsync(o.unit_count[1], 0, (size_t)o.units_cap * 4, s));
    PIE_HIP(c, hipMemsetAsync(o.sum[0], 0, ord_sum_bytes(), s));
    PIE_HIP(c, hipMemsetAsync(o.sum[1], 0, ord_sum_bytes(), s));
    if (m > 0) {
        hipLaunchKernelGGL(k_ord_gather, dim3(c->n_cus * 16), dim3(256), 0, s, built->out_idx, m, built->offsets, o.uoff, c->d_pay, c->d_end,
                           c->key_base, c->key_shift, c->fkey_base, c->fkey_shift, o.pay, o.end, o.key, o.fkey, o.pos);
        PIE_HIP(c, hipGetLastError());
    }
    PIE_HIP(c, hipStreamSynchronize(s));
    o.h_stale[0] = o.h_stale[1] = 0;
    // the build's scan is not a result: the caller's last one (in the other slot) stays readable, and the next scan takes
    // the slot it would have taken
    built->have_result = false;
    c->res = (k.res && k.res != built) ? k.res : nullptr;
    c->next_slot = k.next_slot;
    o.n = total;
    o.held = m;
    o.users = seg_users;
    o.rows = c->n;
    o.uc_next = 0;
    o.valid = true;
    o.wanted = 0;
    o.built_at = c->scans_begun;
    o.builds++;
    timespec t1{};
    clock_gettime(CLOCK_MONOTONIC, &t1);
    o.build_ms = (double)(t1.tv_sec - t0.tv_sec) * 1e3 + (double)(t1.tv_nsec - t0.tv_nsec) * 1e-6;
    return PIE_OK;
}

void launch_ord_append(pie_ctx* c, hipStream_t s, size_t k, long long row0, int n_users, int pass)
{
    OrderedRun& o = c->ord;
    const unsigned grid = (unsigned)((k + 255) / 256);
    const int* st_user = reinterpret_cast<const int*>(c->d_stage + k * 16);
    if (pass == 2) (void)hipMemcpyAsync(o.placed + kOrdAppendMax, o.placed, k * 4, hipMemcpyDeviceToDevice, s);
    hipLaunchKernelGGL(k_ord_append_mark, dim3(grid), dim3(256), 0, s, st_user, (int)k, n_users, o.placed + kOrdAppendMax, pass, o.bhead, o.bnext);
    hipLaunchKernelGGL(k_ord_append, dim3(grid), dim3(256), 0, s, reinterpret_cast<const long long*>(c->d_stage),
                       reinterpret_cast<const long long*>(c->d_stage + k * 8), st_user, reinterpret_cast<const int*>(c->d_stage + k * 20), (int)k,
                       row0, n_users, c->key_base, c->key_shift, c->fkey_base, c->fkey_shift, o.uoff, o.ufill, o.pay, o.end, o.key, o.fkey,
                       o.pos, o.stale, o.placed + kOrdAppendMax, o.placed, pass == 1 ? o.pend : (int*)nullptr, pass, o.bhead, o.bnext);
}

// Segments are full: give every user fresh spare slots (its rows, the rows of this append still waiting, a sixteenth more,
// four) and move the run there — a linear pass, no sort.  Then the waiting rows of the append (still in the staging block)
// take their places.  Returns with the run valid, or invalid when it no longer fits / the second pass failed too.
int ord_respread(pie_ctx* c, size_t k, long long row0, int n_users)
{
    OrderedRun& o = c->ord;
    hipStream_t s = c->stream;
    const size_t padded = (((size_t)o.pos_cap + 1023) / 1024) * 1024 + 1024;
    if (!o.alt_pay) {
        const bool ok = hipMalloc(&o.alt_pay, padded * sizeof(OrdRec)) == hipSuccess && hipMalloc(&o.alt_end, padded * 8) == hipSuccess &&
                        hipMalloc(&o.alt_key, padded * sizeof(lkey_t)) == hipSuccess && hipMalloc(&o.alt_fkey, padded * sizeof(fkey_t)) == hipSuccess &&
                        hipMalloc(&o.alt_uoff, ((size_t)o.cap_users + 1) * 8) == hipSuccess;
        if (!ok) {
            (void)hipGetLastError();
            dfree(o.alt_pay); dfree(o.alt_end); dfree(o.alt_key); dfree(o.alt_fkey); dfree(o.alt_uoff);
            ord_invalidate(c);
            return PIE_OK;
        }
    }
    const int seg_users = o.users;
    std::vector<int> fill((size_t)seg_users + 1), pend((size_t)seg_users + 1);
    PIE_HIP(c, hipMemcpyAsync(fill.data(), o.ufill, (size_t)seg_users * 4, hipMemcpyDeviceToHost, s));
    PIE_HIP(c, hipMemcpyAsync(pend.data(), o.pend, (size_t)seg_users * 4, hipMemcpyDeviceToHost, s));
    PIE_HIP(c, hipMemsetAsync(o.pend, 0, ((size_t)seg_users + 1) * 4, s));
    PIE_HIP(c, hipStreamSynchronize(s));
    std::vector<long long> seg((size_t)seg_users + 1);
    long long total = 0;
    for (int u = 0; u < seg_users; ++u) {
        seg[(size_t)u] = total;
        const long long rows = (long long)fill[(size_t)u] + pend[(size_t)u];
        total += rows + rows / 16 + ord_spare(c);
    }
    seg[(size_t)seg_users] = total;
    if (total > o.pos_cap) { // the table has outgrown what the run's arrays were sized for: a rebuild re-allocates
        ord_free(c);
        return PIE_OK;
    }
    PIE_HIP(c, hipMemcpyAsync(o.alt_uoff, seg.data(), ((size_t)seg_users + 1) * 8, hipMemcpyHostToDevice, s));
    PIE_HIP(c, hipMemsetAsync(o.alt_pay, 0xFF, padded * sizeof(OrdRec), s));
    hipLaunchKernelGGL(k_fill_ll, dim3(c->n_cus * 8), dim3(256), 0, s, o.alt_end, (long long)padded, (long long)INT64_MIN);
    PIE_HIP(c, hipMemsetAsync(o.alt_key, 0, padded * sizeof(lkey_t), s));
    PIE_HIP(c, hipMemsetAsync(o.alt_fkey, 0, padded * sizeof(fkey_t), s));
    hipLaunchKernelGGL(k_ord_respread, dim3(c->n_cus * 16), dim3(256), 0, s, o.n, seg_users, o.uoff, o.ufill, o.alt_uoff, o.pay, o.end, o.key,
                       o.fkey, o.alt_pay, o.alt_end, o.alt_key, o.alt_fkey, o.pos);
    PIE_HIP(c, hipGetLastError());
    std::swap(o.pay, o.alt_pay);
    std::swap(o.end, o.alt_end);
    std::swap(o.key, o.alt_key);
    std::swap(o.fkey, o.alt_fkey);
    std::swap(o.uoff, o.alt_uoff);
    o.n = total;
    o.respreads++;
    o.h_stale[0] = o.h_stale[1] = 0;
    launch_ord_append(c, s, k, row0, n_users, 2);
    PIE_HIP(c, hipGetLastError());
    PIE_HIP(c, hipStreamSynchronize(s));
    if (*(volatile unsigned int*)&o.h_stale[0] || *(volatile unsigned int*)&o.h_stale[1]) {
        o.h_stale[0] = o.h_stale[1] = 0;
        ord_invalidate(c);
    }
    return PIE_OK;
}

long long batch_users_stride(const pie_ctx* c);
long long batch_out_stride(const pie_ctx* c);
size_t batch_ucap(const pie_ctx* c);
size_t batch_mq_bytes();

// the batched form's per-query arrays (allocated at the first ordered batch)
int ord_batch_alloc(pie_ctx* c)
{
    OrderedRun& o = c->ord;
    if (o.bq_local) return PIE_OK;
    const size_t units = (size_t)o.units_cap, groups = units / 1024 + 2;
    const bool ok = hipMalloc(&o.bq_local, units * 4) == hipSuccess && hipMalloc(&o.bq_gsum, groups * 8) == hipSuccess &&
                    hipMalloc(&o.bq_gbase, groups * 8) == hipSuccess && hipMalloc(&o.bq_ctl, sizeof(OrdCtl)) == hipSuccess &&
                    hipMalloc(&o.bq_sum[0], ord_sum_bytes() + batch_mq_bytes()) == hipSuccess &&
                    hipMalloc(&o.bq_sum[1], ord_sum_bytes() + batch_mq_bytes()) == hipSuccess;
    if (!ok) {
        (void)hipGetLastError();
        dfree(o.bq_local); dfree(o.bq_gsum); dfree(o.bq_gbase); dfree(o.bq_ctl); dfree(o.bq_sum[0]); dfree(o.bq_sum[1]);
        return PIE_E_NOMEM;
    }
    PIE_HIP(c, hipMemsetAsync(o.bq_ctl, 0, sizeof(OrdCtl), c->stream));
    PIE_HIP(c, hipMemsetAsync(o.bq_sum[0], 0, ord_sum_bytes() + batch_mq_bytes(), c->stream));
    PIE_HIP(c, hipMemsetAsync(o.bq_sum[1], 0, ord_sum_bytes() + batch_mq_bytes(), c->stream));
    return PIE_OK;
}

// can this batch run on the ordered run?  (a table whose batches the general pass cannot hold: skewed users)
bool ordered_batch_wanted(const pie_ctx* c)
{
    const OrderedRun& o = c->ord;
    if (o.mode == 0 || !o.valid || o.rows != c->n || c->d_qual || !c->key_ok || c->key_poor) return false;
    if (!(o.mode == 2 || c->batch_poor || c->hot_bucket)) return false;
    // union staging records (8 B per position) live in the scan slots' record staging
    const size_t padded = (((size_t)o.pos_cap + 1023) / 1024) * 1024 + 1024;
    return (padded + kOrdTile) * sizeof(OrdUnion) <= (size_t)c->sel_cap * sizeof(SelRec);
}

void fill_batch_tables(const pie_ctx* c, const BatchSlot& b, const pie_query* qs, const unsigned* nk, BatchTables& t);

// a batch on the ordered run (up to 64 queries): scan, ONE prefix, emit, publish — see pie_ordered.h "the UNION form"
void launch_ordered_union(pie_ctx* c, BatchSlot& b, hipStream_t s, const pie_query* qs, bool fine)
{
    OrderedRun& o = c->ord;
    const int bi = (int)(&b - c->bslot) & 1; // scratch (staging, device summary) of two sets: chains run one after the other in the stream
    char* sums = o.bq_sum[bi];
    int* uc = o.unit_count[o.uc_next];
    int* uc_other = o.unit_count[o.uc_next ^ 1];
    o.uc_next ^= 1;
    OrdUnion* ustage = reinterpret_cast<OrdUnion*>(c->slot[bi].sel);
    const unsigned impossible = fine ? 0xFFu : 0xFFFFu; // (a query that falls back is not in the tables)
    unsigned mk = impossible, nk[kBatchMax];
    for (int q = 0; q < b.n_q; ++q) {
        nk[q] = fine ? host_fine_key_of(c, qs[q].now) : host_key_of(c, qs[q].now);
        if (!b.fallback[q] && nk[q] < mk) mk = nk[q];
    }
    BatchTables tab;
    fill_batch_tables(c, b, qs, nk, tab);
    const int chunk_shift = 9;
    const long long n_chunks = (o.n + (1 << chunk_shift) - 1) >> chunk_shift;
    Summary* sum0 = reinterpret_cast<Summary*>(sums);
    unsigned int* mq_slots = reinterpret_cast<unsigned int*>(sums + ord_sum_bytes()); // behind the summary: its own, zero between batches
    if (fine) {
        const int gm = o.grid_mult;
        long long grid = (n_chunks + 3) / 4 < (long long)c->n_cus * gm ? (n_chunks + 3) / 4 : (long long)c->n_cus * gm;
        if (grid < 1) grid = 1;
        hipLaunchKernelGGL((k_ord_batch_scan_t<fkey_t>), dim3((unsigned)grid), dim3(256), 0, s, o.pay, o.end, o.fkey, o.n, n_chunks, mk, tab, ustage, uc, sum0);
    } else {
        long long grid = (n_chunks + 3) / 4 < (long long)c->n_cus * 6 ? (n_chunks + 3) / 4 : (long long)c->n_cus * 6;
        if (grid < 1) grid = 1;
        hipLaunchKernelGGL((k_ord_batch_scan_t<lkey_t>), dim3((unsigned)grid), dim3(256), 0, s, o.pay, o.end, o.key, o.n, n_chunks, mk, tab, ustage, uc, sum0);
    }
    if (b.ev_index >= 0) (void)hipEventRecord(c->ring[b.ev_index].e1, s);
    long long n_groups = (n_chunks + kOrdGroup - 1) >> kOrdGroupShift;
    if (n_groups < 1) n_groups = 1;
    const unsigned pre_grid = (unsigned)(n_groups < (long long)c->n_cus ? n_groups : (long long)c->n_cus);
    hipLaunchKernelGGL(k_ord_prefix, dim3(pre_grid), dim3(256), 0, s, uc, n_chunks, o.bq_local, o.bq_gsum, o.bq_gbase, o.bq_ctl, sum0, 0LL, 0LL, 0LL);
    const int copy_blocks = c->n_cus * 8;
    const int user_blocks = (int)(((long long)c->n_users + 1 + 254) / 255);
    const unsigned egrid = (unsigned)(copy_blocks + user_blocks);
    if (b.n_q > 32)
        hipLaunchKernelGGL((k_ord_union_emit<true>), dim3(egrid), dim3(256), 0, s, o.uoff, c->n_users, o.n, chunk_shift, n_chunks, b.n_q, tab, ustage, uc, o.bq_local,
                           o.bq_gbase, b.uoff, b.urows, b.umlo, b.umhi, (long long)batch_ucap(c), copy_blocks, sum0, mq_slots, uc_other, o.units_cap);
    else
        hipLaunchKernelGGL((k_ord_union_emit<false>), dim3(egrid), dim3(256), 0, s, o.uoff, c->n_users, o.n, chunk_shift, n_chunks, b.n_q, tab, ustage, uc, o.bq_local,
                           o.bq_gbase, b.uoff, b.urows, b.umlo, b.umhi, (long long)batch_ucap(c), copy_blocks, sum0, mq_slots, uc_other, o.units_cap);
    hipLaunchKernelGGL(k_ord_union_publish, dim3(1), dim3(256), 0, s, sum0, mq_slots, b.n_q, (long long)batch_ucap(c), b.bh_dev, b.seq);
}

// should this query run on the ordered run?  (mode 1: where the general path is weak)
bool ordered_wanted(const pie_ctx* c)
{
    if (c->ord.mode == 2) return true;
    if (c->ord.mode == 0) return false;
    return c->hot_bucket || c->batch_poor || (c->last_m >= 0 && c->last_m * 24 > c->n);
}

// keyed (sparse) or dense form: an upper bound on the candidate rows — the key histogram while it describes the
// table, else the live fraction of the last scan
bool ordered_keyed_form(const pie_ctx* c, long long now)
{
    if (c->key_poor) return false;
    double frac = c->live_frac >= 0 ? c->live_frac : 1.0;
    if (!c->key_dirty && c->key_hist_rows == c->n && c->n > 0) {
        frac = (double)rows_keyed_at_or_above(c, now) / (double)c->n;
    }
    return frac < 0.08;
}

// scan kernel, prefix, emit, publish: four launches, nothing to do at pie_scan_finish but read the summary
void launch_ordered(pie_ctx* c, Slot& sl, hipStream_t s, long long now, long long cutoff, unsigned long long mask)
{
    OrderedRun& o = c->ord;
    const int si = (int)(&sl - c->slot);
    Summary* sum = reinterpret_cast<Summary*>(o.sum[si]);
    sl.sum = sum; // what a later pack of this slot's result reads M from
    OrdCtl* ctl = reinterpret_cast<OrdCtl*>(o.sum[si] + span_stats_bytes());
    int* uc = o.unit_count[o.uc_next];
    int* uc_other = o.unit_count[o.uc_next ^ 1];
    o.uc_next ^= 1;
    unsigned int* stage = reinterpret_cast<unsigned int*>(sl.sel);
    const bool keyed = ordered_keyed_form(c, now);
    const bool fine = keyed && (c->k1_keyed & 0x800) && !c->fkey_poor && now >= c->fkey_base;
    long long n_units;
    int unit_shift;
    if (!keyed) {
        unit_shift = kOrdTileShift;
        n_units = (o.n + kOrdTile - 1) >> kOrdTileShift;
        long long grid = n_units < (long long)c->n_cus * 8 ? n_units : (long long)c->n_cus * 8;
        if (grid < 1) grid = 1;
        hipLaunchKernelGGL(k_ord_scan_dense, dim3((unsigned)grid), dim3(256), 0, s, o.pay, o.end, o.key, o.n, now, host_key_of(c, now),
                           cutoff, mask, stage, uc, o.tile_ballot, o.tile_prefix, sum);
        sl.variant = 0x2003;
    } else if (fine) {
        unit_shift = 9;
        n_units = (o.n + 511) >> 9;
        const int gm = o.grid_mult; // blocks per CU (profiles/r02_ze_ord_keyed_grid_unroll_sweep.txt: 12 beats 4 .. 10; the unroll does not matter)
        long long grid = (n_units + 3) / 4 < (long long)c->n_cus * gm ? (n_units + 3) / 4 : (long long)c->n_cus * gm;
        if (grid < 1) grid = 1;
        hipLaunchKernelGGL((k_ord_scan_keyed<fkey_t>), dim3((unsigned)grid), dim3(256), 0, s, o.pay, o.end, o.fkey, o.n, n_units, now,
                           host_fine_key_of(c, now), cutoff, mask, stage, uc, sum);
        sl.variant = 0x2C00;
    } else {
        unit_shift = 9;
        n_units = (o.n + 511) >> 9;
        long long grid = (n_units + 3) / 4 < (long long)c->n_cus * 8 ? (n_units + 3) / 4 : (long long)c->n_cus * 8;
        if (grid < 1) grid = 1;
        hipLaunchKernelGGL((k_ord_scan_keyed<lkey_t>), dim3((unsigned)grid), dim3(256), 0, s, o.pay, o.end, o.key, o.n, n_units, now,
                           host_key_of(c, now), cutoff, mask, stage, uc, sum);
        sl.variant = 0x2400;
    }
    sl.k1_blocks = 0;
    if (sl.ev_index >= 0) (void)hipEventRecord(c->ring[sl.ev_index].e1, s);
    long long n_groups = (n_units + kOrdGroup - 1) >> kOrdGroupShift;
    if (n_groups < 1) n_groups = 1;
    const unsigned pre_grid = (unsigned)(n_groups < (long long)c->n_cus * 4 ? n_groups : (long long)c->n_cus * 4);
    hipLaunchKernelGGL(k_ord_prefix, dim3(pre_grid), dim3(256), 0, s, uc, n_units, o.unit_local, o.group_sum, o.group_base, ctl, sum, 0LL, 0LL, 0LL);
    const int fin_blocks = (int)(((long long)c->n_users + 1 + 254) / 255);
    if (keyed) {
        const int copy_blocks = c->n_cus * 8;
        hipLaunchKernelGGL(k_ord_emit<true>, dim3((unsigned)(copy_blocks + fin_blocks)), dim3(256), 0, s, o.uoff, c->n_users, o.n, unit_shift,
                           uc, o.unit_local, o.group_base, n_units, stage, o.pay, o.tile_ballot, o.tile_prefix, sl.out_idx, sl.offsets,
                           sl.counts_ord, copy_blocks, sum, uc_other, o.units_cap);
    } else {
        long long copy_blocks = n_units < (long long)c->n_cus * 16 ? n_units : (long long)c->n_cus * 16;
        if (copy_blocks < 1) copy_blocks = 1;
        hipLaunchKernelGGL(k_ord_emit<false>, dim3((unsigned)(copy_blocks + fin_blocks)), dim3(256), 0, s, o.uoff, c->n_users, o.n, unit_shift,
                           uc, o.unit_local, o.group_base, n_units, stage, o.pay, o.tile_ballot, o.tile_prefix, sl.out_idx, sl.offsets,
                           sl.counts_ord, (int)copy_blocks, sum, uc_other, o.units_cap);
    }
    hipLaunchKernelGGL(k_ord_publish, dim3(1), dim3(64), 0, s, sum, sl.h_sum_dev, sl.seq, 0LL);
}

int scan_begin(pie_ctx* c, long long now, long long cutoff, int* msg = nullptr, int msg_u_pad = 0, long long msg_cap = 0,
               int* msg_counts = nullptr);
int scan_begin(pie_ctx* c, long long now, long long cutoff, int* msg, int msg_u_pad, long long msg_cap, int* msg_counts)
{
    if (c->cap_rows == 0) return fail(c, PIE_E_STATE, "no table loaded");
    if (c->n_flight >= 2) return fail(c, PIE_E_STATE, "two scans are already in flight: call pie_scan_finish first");
    if (c->key_rebuild && c->n_flight == 0) {
        int rc = build_keys(c, 0, true);
        if (rc) return rc;
    }
    if (c->dshift_want > c->dshift && c->n_flight == 0 && c->slot[0].direct) {
        // larger direct slots for every user: both slots' arrays are replaced (their contents are per-scan)
        int rc = sync_all(c);
        if (rc) return rc;
        BktRec* fresh[2] = {nullptr, nullptr};
        const size_t bytes = ((size_t)c->cap_users << c->dshift_want) * sizeof(BktRec);
        if (hipMalloc(&fresh[0], bytes) == hipSuccess && hipMalloc(&fresh[1], bytes) == hipSuccess) {
            for (int k = 0; k < 2; ++k) {
                (void)hipFree(c->slot[k].direct);   // scratch of finished scans only: their results live in out_idx / offsets / counts
                c->slot[k].direct = fresh[k];
            }
            c->dshift = c->dshift_want;
        } else {
            (void)hipGetLastError();
            if (fresh[0]) (void)hipFree(fresh[0]);
            c->dshift_want = c->dshift; // no room: stay with what there is
        }
    }
    // The ordered run serves the queries the general path is weak at (dense, skewed users).  It is built the second time
    // in a row such a query arrives with nothing in flight and the table unchanged (a first dense query may be the only one).
    bool ord_route = false;
    if (!c->d_qual && !c->k1_pinned && c->key_ok && !c->ord.no_room && ordered_wanted(c)) {
        if (!c->ord.valid && c->n_flight <= 1 && (c->ord.mode == 2 || ++c->ord.wanted >= c->ord.need)) {
            int rc = build_ordered(c);
            if (rc) return rc;
        }
        ord_route = c->ord.valid && c->ord.rows == c->n;
    } else c->ord.wanted = 0;
    Slot& sl = c->slot[c->next_slot];
    hipStream_t s = c->stream;
    sl.ev_index = -1;
    if (c->profiling && (c->scans_begun % (unsigned long long)c->profile_every) == 0) {
        if (c->ring_used == kEventRing) {
            int rc = resolve_events(c);
            if (rc) return rc;
        }
        if (c->ring_used < kEventRing) {
            if ((int)c->ring.size() <= c->ring_used) {
                ScanEvents e{};
                PIE_HIP(c, hipEventCreate(&e.e0));
                PIE_HIP(c, hipEventCreate(&e.e1));
                PIE_HIP(c, hipEventCreate(&e.e2));
                c->ring.push_back(e);
            }
            sl.ev_index = c->ring_used++;
        }
    }
    const unsigned long long mask = c->n_disc >= 64 ? c->disc_mask : (c->disc_mask & ((1ull << c->n_disc) - 1ull));
    sl.variant = c->k1_variant;
    if (c->d_qual) sl.variant = 0x101;
    else if (!c->k1_pinned && c->live_frac >= 0) {
        sl.variant = c->live_frac < kLiveFirstBelow ? c->k1_live_first : c->k1_variant;
        // few candidates: stream the 2-byte liveness key instead of the 8-byte `end` column
        if ((sl.variant & 4) && c->keyed_enabled && c->key_ok && !c->key_poor) {
            sl.variant = c->k1_keyed & ~0x800;
            // queries above the fine key's base (every query that few rows survive) stream 1 B/row instead of 2
            if ((c->k1_keyed & 0x800) && !c->fkey_poor && now >= c->fkey_base) sl.variant |= 0x800;
        }
    }
    else if (!c->k1_pinned && !c->d_qual && c->live_frac < 0 && c->keyed_enabled && c->key_ok && !c->key_dirty &&
             c->key_hist_rows == c->n && c->n > 0) {
        // first scan of a table: no scan has counted live rows yet, but the key histogram taken when the key columns were
        // built bounds them: rows whose key bin lies at or above the bin of key(now) (an upper bound on the live rows)
        if ((double)rows_keyed_at_or_above(c, now) < kLiveFirstBelow * (double)c->n) {
            sl.variant = c->k1_keyed & ~0x800;
            if ((c->k1_keyed & 0x800) && now >= c->fkey_base) sl.variant |= 0x800;
        }
    }
    if ((sl.variant & 0x400) && !c->key_ok) sl.variant = c->k1_live_first; // pinned keyed form without a key column
    // rows clustered by user + a dense query: the streaming form aggregates its histogram atomics per wave
    if (!c->k1_pinned && !c->d_qual && sl.variant == 0x03 && c->clustered) sl.variant = 0x43;
    // skewed users (one bucket held > 1/64 of the last scan's selected rows): aggregate the histogram atomics per wave
    if (!c->k1_pinned && !c->d_qual && (sl.variant & 4) && c->hot_bucket) sl.variant |= 0x40;
    sl.hot.n = 0;
    if ((sl.variant & 0x44) == 0x44 && !c->d_qual) sl.hot = c->hot; // aggregated forms only; K1 and K3 of this scan share it
    const int plan = (sl.variant & 0x800) ? 3 : (sl.variant & 0x400) ? 2 : (sl.variant & 4) ? 1 : 0;
    sl.k1_blocks = c->plan_blocks[plan];
    sl.rows_per_block = c->plan_rows[plan];
    sl.have_result = false;
    if (c->res == &sl) c->res = nullptr;
    sl.msg = msg;
    sl.msg_u_pad = msg_u_pad;
    sl.msg_cap = msg_cap;
    sl.msg_counts = msg_counts;
    sl.msg_by_k2 = false;
    sl.ordered = false;

    if (ord_route) {
        {
            Slot& prev = c->slot[c->next_slot ^ 1];
            if (c->n_flight == 1 && prev.in_flight && prev.k2_pending) { // nothing carries it along: it goes first
                launch_k2(c, prev, s, prev.zero_span, (long long)(counts_span(c) / 16));
                prev.k2_pending = false;
            }
            sl.ordered = true;
            sl.fast = false;
            sl.k2_pending = false;
            sl.hot.n = 0;
            c->scans_begun++;
            sl.q_now = now;
            sl.q_cutoff = cutoff;
            sl.seq = ++c->seq_counter;
            if (sl.ev_index >= 0) PIE_HIP(c, hipEventRecord(c->ring[sl.ev_index].e0, s));
            launch_ordered(c, sl, s, now, cutoff, mask);
            PIE_HIP(c, hipGetLastError());
            if (sl.ev_index >= 0) PIE_HIP(c, hipEventRecord(c->ring[sl.ev_index].e2, s));
            sl.in_flight = true;
            c->n_flight++;
            c->next_slot ^= 1;
            return PIE_OK;
        }
    }

    // this scan's histogram span (zeroed by the previous scan's K2, or by the load) and the one K2 will zero
    {
        char* base = c->span[c->span_next];
        sl.counts = reinterpret_cast<int*>(base);
        sl.tile_pub = reinterpret_cast<unsigned long long*>(base + span_counts_bytes(c));
        sl.part_cursor = reinterpret_cast<int*>(base + span_counts_bytes(c) + span_tiles_bytes(c));
        sl.ctl = reinterpret_cast<ScanCtl*>(base + span_counts_bytes(c) + span_tiles_bytes(c) + span_parts_bytes());
        sl.sum = reinterpret_cast<Summary*>(base + span_counts_bytes(c) + span_tiles_bytes(c) + span_parts_bytes() + 128);
        c->span_next = (c->span_next + 1) % 3;
    }
    // K2 (or the fast path's tail) zeroes the span of the scan AFTER the next one: every span is clean again before its
    // next user starts, and K2 of this scan may run beside the next scan's table pass, which uses span_next
    int4* zero_span = reinterpret_cast<int4*>(c->span[(c->span_next + 1) % 3]);
    if (c->ord_building) zero_span = nullptr; // see build_ordered: this scan stands outside the rotation
    sl.zero_span = zero_span;
    c->scans_begun++;
    sl.q_now = now;
    sl.q_cutoff = cutoff;
    sl.seq = ++c->seq_counter;
    // Partitioned fast path: the query is sparse (liveness-first form chosen), the previous scan's M says a user
    // range of 1 << part_shift users holds far fewer than kPartCap selected rows, and no partition ever overflowed
    // on this table.  Two launches, no host round trip: K1P, then the one-wave-per-partition tail.
    sl.fast = false;
    if (!c->k1_pinned && !c->d_qual && c->fast_enabled && (sl.variant & 4) && c->part_shift >= 0 && c->last_m >= 0) {
        const double mean = (double)c->last_m / (double)c->n_parts;
        sl.fast = mean + 6.0 * __builtin_sqrt(mean) + 16.0 <= (double)kPartCap;
    }
    if (sl.fast) {
        // the fast path's tail zeroes the span the scan in flight still needs for its K2: that K2 goes first
        Slot& prev = c->slot[c->next_slot ^ 1];
        if (c->n_flight == 1 && prev.in_flight && prev.k2_pending) {
            launch_k2(c, prev, s, prev.zero_span, (long long)(counts_span(c) / 16));
            prev.k2_pending = false;
        }
        sl.variant = 0x285;
        sl.k1_blocks = c->plan_blocks[1];
        sl.rows_per_block = c->plan_rows[1];
        if (sl.ev_index >= 0) PIE_HIP(c, hipEventRecord(c->ring[sl.ev_index].e0, s));
        hipLaunchKernelGGL((k_scan_live_first_part<8, true>), dim3(sl.k1_blocks), dim3(kK1Threads), 0, s, c->d_start, c->d_end,
                           c->d_user, c->d_disc, c->n, sl.rows_per_block, now, cutoff, mask, c->n_users, c->part_shift,
                           sl.part_cursor, sl.part_rec, sl.sum);
        if (sl.ev_index >= 0) PIE_HIP(c, hipEventRecord(c->ring[sl.ev_index].e1, s));
        const unsigned tail_waves = kTailThreads / 64;
        unsigned tail_blocks = (unsigned)((c->n_parts + tail_waves - 1) / tail_waves);
        if (tail_blocks > (unsigned)c->n_cus) tail_blocks = (unsigned)c->n_cus;
        hipLaunchKernelGGL(k_tail_partitions, dim3(tail_blocks), dim3(kTailThreads), 0, s, sl.part_cursor, sl.part_rec, c->n_parts,
                           c->part_shift, c->n_users, sl.counts_ord, sl.offsets, sl.out_idx, sl.sum,
                           reinterpret_cast<unsigned int*>(sl.blk_count), zero_span, (long long)(counts_span(c) / 16));
        hipLaunchKernelGGL(k_publish_summary, dim3(1), dim3(256), 0, s, sl.sum, reinterpret_cast<unsigned int*>(sl.blk_count),
                           (int)tail_blocks, sl.h_sum_dev, sl.seq);
        PIE_HIP(c, hipGetLastError());
        if (sl.ev_index >= 0) PIE_HIP(c, hipEventRecord(c->ring[sl.ev_index].e2, s));
        sl.in_flight = true;
        c->n_flight++;
        c->next_slot ^= 1;
        return PIE_OK;
    }
    // K2 of the scan already in flight (if it is still pending) rides in this scan's launch when both fit the fused
    // kernel; otherwise it goes first, on its own
    Slot& other = c->slot[c->next_slot ^ 1];
    const bool ride = c->n_flight == 1 && other.in_flight && other.k2_pending && tail_can_ride(c, other) &&
                      (keyed_can_carry(c, sl) || stream_can_carry(c, sl));
    if (c->n_flight == 1 && other.in_flight && other.k2_pending && !ride) {
        launch_k2(c, other, s, other.zero_span, (long long)(counts_span(c) / 16));
        other.k2_pending = false;
    }
    if (sl.ev_index >= 0) PIE_HIP(c, hipEventRecord(c->ring[sl.ev_index].e0, s));
    if (ride) {
        launch_keyed_with_tail(c, sl, other, s, now, cutoff, mask);
        other.k2_pending = false;
    } else {
        launch_k1(c, sl, s, now, cutoff, mask);
    }
    if (sl.ev_index >= 0) PIE_HIP(c, hipEventRecord(c->ring[sl.ev_index].e1, s));
    sl.k2_pending = true; // launched by the next pie_scan_begin (riding) or by pie_scan_finish, whichever comes first
    PIE_HIP(c, hipGetLastError());
    sl.in_flight = true;
    c->n_flight++;
    c->next_slot ^= 1;
    return PIE_OK;
}

// Chunk interleave of the next keyed / batched pass, from what the last one saw: `chunk_max` = candidates in its densest
// chunk (1024 or 512 rows), `cand` = all candidates.  Live rows spread evenly (the synthetic corpus of SURVEY.md 8d: a
// chunk holds a handful) -> eight consecutive chunks per wave; a chunk that is an eighth full while the average chunk
// is nearly empty (a session table in creation order: every live row at its end; a login burst) -> fully interleaved.
void choose_run_shift(pie_ctx* c, unsigned long long cand, unsigned chunk_max, bool fine_key)
{
    if (c->run_shift_pinned) return;
    const double rows_per_chunk = fine_key ? (double)kFineKeyRowsPerLoad : (double)kKeyRowsPerLoad;
    const double chunks = (double)c->n / rows_per_chunk;
    if (chunks < 1.0) return;
    const double mean = (double)cand / chunks;
    const bool dense_stretch = (double)chunk_max >= rows_per_chunk / 8.0 && (double)chunk_max > 8.0 * (mean + 1.0);
    c->run_shift = dense_stretch ? 0 : 3;
}

// Tail of the oldest scan in flight: launch its K2 if no later scan took it along, wait for its summary, then — only for
// buckets that outgrew their direct slots — scatter + per-bucket order (plus the merge passes of big buckets, sized from
// the summary).  Also where the adaptive choices for the next scans are made (scan form, key fit, hot set, slot capacity).
// `which`: finish THIS slot's scan although an older one is still in flight (the ordered run's build, which must not
// consume the caller's scan); everything a finish launches works on its own slot's arrays, in stream order
int scan_finish(pie_ctx* c, Slot* which)
{
    Slot* slp = which ? which : oldest_in_flight(c);
    if (!slp) return fail(c, PIE_E_STATE, "pie_scan_finish without pie_scan_begin");
    Slot& sl = *slp;
    hipStream_t a = c->stream;
    const bool e2_recorded = sl.k2_pending && sl.ev_index >= 0;
    if (sl.k2_pending) { // no later scan took it along
        launch_k2(c, sl, a, sl.zero_span, (long long)(counts_span(c) / 16));
        PIE_HIP(c, hipGetLastError());
        sl.k2_pending = false;
        // "last kernel end" of a scan whose buckets all fit their slots is the end of K2: the event goes in right behind it,
        // not after the host has woken up on the summary (a scan that needs K3 / K4 records it again behind those)
        if (sl.ev_index >= 0) PIE_HIP(c, hipEventRecord(c->ring[sl.ev_index].e2, a));
    }
    // wait for K2's last block to publish the summary in mapped host memory (no copy node, no event wait).  The wait is
    // bounded (PIE_WAIT_DEADLINE_MS, default 20 s): a kernel that never finishes must not hang the caller — in the Node
    // host that is the JS main thread.  The slot stays in flight until its summary has been read, so an error return
    // leaves the context consistent: the failed scan is dropped (no result), the stream is left to drain.
    {
        volatile unsigned long long* seq = &sl.h_sum->seq;
        unsigned long long spins = 0;
        bool from_device = false;
        timespec t0{};
        clock_gettime(CLOCK_MONOTONIC, &t0);
        while (*seq != sl.seq) {
            __builtin_ia32_pause();
            if ((++spins & 0x3FFFF) == 0) { // every ~1 ms: is the stream still healthy, is the deadline up?
                hipError_t q = hipStreamQuery(c->stream);
                if (q == hipSuccess && *seq != sl.seq && !sl.ordered) {
                    // the stream drained but the mapped write is not visible: read the device copy instead
                    from_device = true;
                    break;
                }
                timespec t1{};
                clock_gettime(CLOCK_MONOTONIC, &t1);
                const double waited_ms = (double)(t1.tv_sec - t0.tv_sec) * 1e3 + (double)(t1.tv_nsec - t0.tv_nsec) * 1e-6;
                if ((q != hipSuccess && q != hipErrorNotReady) || waited_ms > c->wait_deadline_ms) {
                    sl.in_flight = false;
                    sl.have_result = false;
                    c->n_flight--;
                    if (c->res == &sl) c->res = nullptr;
                    if (q != hipSuccess && q != hipErrorNotReady) return fail(c, PIE_E_HIP, "scan failed: %s", hipGetErrorString(q));
                    return fail(c, PIE_E_HIP, "scan summary not published within %.0f ms (PIE_WAIT_DEADLINE_MS): kernel hung?", c->wait_deadline_ms);
                }
            }
        }
        __atomic_thread_fence(__ATOMIC_ACQUIRE);
        if (from_device) {
            // K2 writes live / amb only to the host copy: take every other field from the device Summary and the row
            // statistics from their slots (the same sum K2's last block makes)
            std::vector<unsigned char> raw(span_stats_bytes());
            hipError_t e = hipMemcpy(raw.data(), sl.sum, raw.size(), hipMemcpyDeviceToHost);
            if (e != hipSuccess) {
                sl.in_flight = false;
                c->n_flight--;
                return fail(c, PIE_E_HIP, "summary read-back: %s", hipGetErrorString(e));
            }
            Summary dev;
            memcpy(&dev, raw.data(), sizeof dev);
            dev.live = dev.amb = dev.cand = 0;
            dev.chunk_max = 0;
            for (int k = 0; k < kStatSlots; ++k) {
                StatSlot st;
                memcpy(&st, raw.data() + kSummaryBytes + (size_t)k * sizeof(StatSlot), sizeof st);
                dev.live += st.live;
                dev.amb += st.amb;
                dev.cand += st.cand;
                dev.chunk_max = st.chunk_max > dev.chunk_max ? (unsigned)st.chunk_max : dev.chunk_max;
            }
            sl.h_sum->s = dev;
        }
    }
    sl.in_flight = false;
    c->n_flight--;
    sl.last = sl.h_sum->s;
    if (sl.ordered) {
        // idx, offsets and counts are complete; what the scan saw of the table steers the next one like any other scan's
        c->last_m = (long long)sl.last.m;
        c->live_frac = c->ord.n > 0 ? (double)sl.last.live / (double)c->n : 0.0;
        c->hot_bucket = sl.last.m > 4096 && (unsigned long long)sl.last.max_count * 64ull > sl.last.m;
        if ((sl.variant & 0x400) && sl.last.amb > 4096 && sl.last.amb > (unsigned long long)c->n / 64) {
            if (c->key_dirty) c->key_rebuild = true;
            else if (sl.variant & 0x800) c->fkey_poor = true;
            else c->key_poor = true;
        }
        sl.have_result = true;
        c->res = &sl;
        c->last_was_batch = false;
        return PIE_OK;
    }
    if (sl.fast) {
        if (sl.last.pad & 1u) {
            // a partition overflowed (skewed users): everything this attempt wrote is discarded and the same query
            // reruns on the general path, here and now; the fast path stays off for this table
            c->fast_enabled = false;
            int rc = sync_all(c);
            if (rc) return rc;
            const unsigned long long mask = c->n_disc >= 64 ? c->disc_mask : (c->disc_mask & ((1ull << c->n_disc) - 1ull));
            PIE_HIP(c, hipMemsetAsync(sl.counts, 0, counts_span(c), a));
            sl.fast = false;
            sl.variant = c->k1_live_first;
            sl.ev_index = -1; // the events of the discarded attempt stay as they are (they timed real launches)
            sl.seq = ++c->seq_counter;
            launch_k1(c, sl, a, sl.q_now, sl.q_cutoff, mask);
            launch_k2(c, sl, a, (int4*)nullptr, 0LL);
            PIE_HIP(c, hipGetLastError());
            PIE_HIP(c, hipStreamSynchronize(a));
            sl.last = sl.h_sum->s;
        } else {
            // nothing left to launch: the tail ran right behind the table pass
            c->live_frac = c->n > 0 ? (double)sl.last.live / (double)c->n : 0.0;
            c->hot_bucket = sl.last.m > 4096 && (unsigned long long)sl.last.max_count * 64ull > sl.last.m;
            c->last_m = (long long)sl.last.m;
            sl.have_result = true;
            c->res = &sl;
            if (sl.last.bad_rows)
                return fail(c, PIE_E_INVAL, "%u selected rows carry a user id outside [0, %d)", sl.last.bad_rows, c->n_users);
            return PIE_OK;
        }
    }
    if ((sl.variant & 0x400) && !c->d_qual) choose_run_shift(c, sl.last.cand, sl.last.chunk_max, (sl.variant & 0x800) != 0);
    if ((sl.variant & 0x400) && sl.last.amb > 4096 && sl.last.amb > (unsigned long long)c->n / 64) {
        // the key column separated this query badly (e.g. `now` beyond the range it was built for)
        if (c->key_dirty) c->key_rebuild = true;
        else if (sl.variant & 0x800) c->fkey_poor = true;
        else c->key_poor = true;
    }
    if (!c->d_qual) {
        c->last_m = (long long)sl.last.m;
        c->live_frac = c->n > 0 ? (double)sl.last.live / (double)c->n : 0.0;
        c->hot_bucket = sl.last.m > 4096 && (unsigned long long)sl.last.max_count * 64ull > sl.last.m;
        // buckets outgrew the direct slots: ask for a capacity that holds the largest one (up to what one wave orders),
        // within the memory budget; applied at the next begin with nothing in flight
        if (sl.direct && sl.last.n_over > 0 && sl.last.max_count > (1u << c->dshift)) {
            int want = c->dshift;
            while (want < 9 && (1u << want) < sl.last.max_count) ++want;
            while (want > c->dshift && ((size_t)c->cap_users << want) * sizeof(BktRec) > kDirectMaxBytes) --want;
            if (want > c->dshift_want) c->dshift_want = want;
        }
        // streaming forms report how many selected rows sat next to (0x03) / shared an atomic with (0x43) a row of the same user
        if (sl.variant == 0x03 || sl.variant == 0x43) c->clustered = sl.last.m > 4096 && sl.last.amb * 2 > sl.last.m;
        // hot set = the users K2 reported (bucket > 1/256 of the previous M), re-read only when their number changed
        // or every 64 scans (the read waits for the stream, so it must stay rare); a stale set is still exact
        const unsigned n_hot = sl.last.n_hot < (unsigned)kHotMax ? sl.last.n_hot : (unsigned)kHotMax;
        c->hot_age++;
        if (n_hot == 0) { c->hot.n = 0; c->hot_seen = 0; }
        else if (n_hot != c->hot_seen || c->hot_age >= 64) {
            PIE_HIP(c, hipMemcpyAsync(c->hot.user, sl.hot_list, (size_t)n_hot * 4, hipMemcpyDeviceToHost, a));
            PIE_HIP(c, hipStreamSynchronize(a));
            c->hot.n = (int)n_hot;
            c->hot_seen = n_hot;
            c->hot_age = 0;
        }
    }

    // Without direct slots every record was staged: scatter, then order the tiny buckets.  With them only the buckets K2
    // counted in n_over (outgrew the slot capacity, or belong to a hot user) have staged records.
    const bool staged = sl.direct ? (sl.last.n_over > 0) : (sl.last.m > 0);
    if (sl.last.m > 0) {
        if (staged) {
            int scat_blocks = sl.k1_blocks;
            if (scat_blocks > c->n_cus * 16) scat_blocks = c->n_cus * 16;
            hipLaunchKernelGGL(k_scatter, dim3(scat_blocks), dim3(256), 0, a, sl.sel, sl.sel_rank, sl.blk_count, sl.k1_blocks,
                               sl.rows_per_block, sl.offsets, sl.bkt, sl.hot, sl.blk_hot_base);
            if (sl.direct) {
                unsigned cp_blocks = (sl.last.n_over + 3) / 4;
                if (cp_blocks > (unsigned)c->n_cus * 8) cp_blocks = (unsigned)c->n_cus * 8;
                hipLaunchKernelGGL(k_copy_direct, dim3(cp_blocks), dim3(256), 0, a, sl.over_list, sl.sum, direct_of(c, sl), sl.offsets, sl.bkt);
            }
        }
        if (!sl.direct) launch_sort_tiny(c, sl, a);
        if (sl.last.n_seg > 0) {
            const unsigned seg_blocks = sl.last.n_seg < (unsigned)(c->n_cus * 3) ? sl.last.n_seg : (unsigned)(c->n_cus * 3);
            hipLaunchKernelGGL(k_sort_segments, dim3(seg_blocks), dim3(1024), 0, a, sl.seg_list, sl.sum, sl.bkt, sl.out_idx);
        }
        if (sl.last.n_small > 0) {
            unsigned small_blocks = (sl.last.n_small + 3) / 4;
            if (small_blocks > (unsigned)c->n_cus * 8) small_blocks = (unsigned)c->n_cus * 8;
            hipLaunchKernelGGL(k_sort_small, dim3(small_blocks), dim3(256), 0, a, sl.small_list, sl.sum, sl.bkt, direct_of(c, sl), sl.out_idx);
        }
    }
    if (sl.last.n_big > 0) {
        // big buckets: tiles of kSegMax were sorted in place by K4; merge passes ping-pong between the bucket
        // arrays and scratch carved out of this slot's (now consumed) record staging; the last pass lands in out_idx.
        BktRec* tmp = reinterpret_cast<BktRec*>(sl.sel); // same 16-B records, n of them
        // up to kMergeWays-way passes (one launch merges a bucket of up to 16 tiles); every pass takes as many ways as it
        // needs to finish, at most kMergeWays
        int passes = 0;
        for (long long w = kSegMax; w < (long long)sl.last.max_count; w *= kMergeWays) ++passes;
        bool in_bkt = true; // which buffer holds the current runs
        long long w = kSegMax;
        for (int p = 0; p < passes; ++p) {
            long long need = ((long long)sl.last.max_count + w - 1) / w; // runs of the largest bucket
            const int ways = need < 2 ? 2 : (need > kMergeWays ? kMergeWays : (int)need);
            const BktRec* src = in_bkt ? sl.bkt : tmp;
            BktRec* dst = in_bkt ? tmp : sl.bkt;
            int* idx_only = (p == passes - 1) ? sl.out_idx : nullptr;
            const unsigned gx = (unsigned)((sl.last.max_count + 255u) / 256u);
            const unsigned gy = sl.last.n_big < 65535u ? sl.last.n_big : 65535u;
            hipLaunchKernelGGL(k_merge_pass, dim3(gx < 4096u ? gx : 4096u, gy), dim3(256), 0, a, sl.big_list,
                               (int)sl.last.n_big, sl.counts_ord, sl.offsets, w, ways, src, dst, idx_only);
            in_bkt = !in_bkt;
            w *= ways;
        }
    }
    PIE_HIP(c, hipGetLastError());
    // kernels were queued behind K2 (or K2 rode in a later scan's launch: the event has not been recorded at all)
    const bool more = (sl.last.m > 0 && (staged || !sl.direct || sl.last.n_seg > 0 || sl.last.n_small > 0)) || sl.last.n_big > 0;
    if (sl.ev_index >= 0 && (more || !e2_recorded)) PIE_HIP(c, hipEventRecord(c->ring[sl.ev_index].e2, a));
    sl.have_result = true;
    c->res = &sl;
    c->last_was_batch = false;
    if (sl.last.bad_rows)
        return fail(c, PIE_E_INVAL, "%u selected rows carry a user id outside [0, %d)", sl.last.bad_rows, c->n_users);
    return PIE_OK;
}

int run_scan(pie_ctx* c, long long now, long long cutoff)
{
    if (c->n_flight) return fail(c, PIE_E_STATE, "a scan is in flight: finish it first");
    int rc = scan_begin(c, now, cutoff);
    if (rc) return rc;
    return scan_finish(c);
}

// ordered list of the rows matching a one-column predicate (expired queue / user match), through slot 0's
// workspace: blk_count + blk_off for the per-block prefix, out_idx as the device-side list
template <int MODE>
int run_row_list(pie_ctx* c, long long a, long long b, int32_t* out, size_t cap, size_t* k_out)
{
    if (k_out) *k_out = 0;
    if (c->cap_rows == 0) return fail(c, PIE_E_STATE, "no table loaded");
    if (c->n == 0) return PIE_OK;
    if (c->n_flight || c->b_flight) return fail(c, PIE_E_STATE, "a scan is in flight");
    PIE_HIP(c, hipSetDevice(c->device));
    int rc = sync_all(c); // the workspace below is shared with the scans' tails
    if (rc) return rc;
    hipStream_t s = c->stream;
    Slot& sl = c->slot[0];
    sl.have_result = false;
    if (c->res == &sl) c->res = nullptr;
    const int blocks = c->plan_blocks[0];
    const long long rpb = c->plan_rows[0];
    // MODE 2 matches on the start column: it travels through the kernels' generic second-column pointer
    const int* col2 = (MODE == 2 || MODE == 3 || MODE == 4) ? reinterpret_cast<const int*>(c->d_start) : c->d_user;
    hipLaunchKernelGGL(k_list_count<MODE>, dim3(blocks), dim3(256), 0, s, c->d_end, col2, c->n, rpb, a, b, sl.blk_count);
    hipLaunchKernelGGL(k_block_prefix, dim3(1), dim3(256), 0, s, sl.blk_count, blocks, c->d_blk_off, &c->d_summary->m);
    hipLaunchKernelGGL(k_list_write<MODE>, dim3(blocks), dim3(256), 0, s, c->d_end, col2, c->n, rpb, a, b,
                       c->d_blk_off, sl.out_idx, c->cap_rows, c->d_key, c->d_fkey, ord_mirror_of(c));
    PIE_HIP(c, hipGetLastError());
    PIE_HIP(c, hipMemcpyAsync(c->h_summary, c->d_summary, sizeof(Summary), hipMemcpyDeviceToHost, s));
    PIE_HIP(c, hipStreamSynchronize(s));
    const size_t k = (size_t)c->h_summary->m;
    if (k_out) *k_out = k;
    if (out && k > cap) return fail(c, PIE_E_CAPACITY, "list cap %zu < %zu", cap, k);
    if (out && k) {
        PIE_HIP(c, hipMemcpyAsync(out, sl.out_idx, k * 4, hipMemcpyDeviceToHost, s));
        PIE_HIP(c, hipStreamSynchronize(s));
    }
    return PIE_OK;
}


// ---------------------------------------------------------------------------------------------- batched scans
// Q <= 64 queries in one table pass (pie_kernels.h "batched scan").  The pass and its tail produce the UNION of the selections
// (per user the rows any query selected, ordered, with a query mask per row): that is the batch's result.  Per-query counts /
// offsets / row lists live in list storage that is allocated and filled only when somebody asks for them
// (pie_batch_read_results, per-query messages) or when a path produces them anyway (fallback scans, the ordered run).

long long batch_users_stride(const pie_ctx* c) { return (((long long)c->cap_users + 1 + 31) / 32) * 32; }
long long batch_out_stride(const pie_ctx* c) { return (long long)c->cap_users * kTinyMax; }
size_t batch_mq_bytes() { return (size_t)kMqSlots * kBatchMax * 4; }
size_t batch_span_bytes(const pie_ctx* c) { return counts_span(c) + batch_mq_bytes(); }

bool batch_supported(const pie_ctx* c)
{
    const int tiles = (c->n_users + kK1Threads - 1) / kK1Threads;
    return c->key_ok && c->d_key && c->d_fkey && c->d_pay && c->slot[0].direct != nullptr && tiles <= kOrderMaxTiles && c->n > 0 &&
           (long long)c->cap_users * kTinyMax < (1LL << 31);
}

// the arrays of the batched pass: union bucket slots + the union result per batch slot, three rotating spans
// entries of the union result arrays: the bucket slots of the general pass, or a sixteenth of the rows (a batch on the ordered
// run has no slot bound: a head user's union is as long as its live rows)
size_t batch_ucap(const pie_ctx* c)
{
    const size_t slots = (size_t)c->cap_users << c->bdshift, rows16 = (size_t)c->cap_rows / 16 + 4096;
    return slots > rows16 ? slots : rows16;
}

bool any_lane_alloc(const pie_ctx* c)
{
    for (bool a : c->lane_alloc)
        if (a) return true;
    return false;
}

hipStream_t lane_stream_of(const pie_ctx* c, int lane) { return lane == 0 ? c->stream : c->lane_stream[lane]; }

// the slot arrays and spans of one lane (and, for lanes above 0, its stream)
int ensure_batch(pie_ctx* c, int lane)
{
    if (c->lane_alloc[lane]) return PIE_OK;
    if (lane > 0 && !c->lane_stream[lane]) PIE_HIP(c, hipStreamCreateWithFlags(&c->lane_stream[lane], hipStreamNonBlocking));
    for (hipEvent_t& e : c->lane_event[lane])
        if (!e) PIE_HIP(c, hipEventCreateWithFlags(&e, hipEventDisableTiming));
    const size_t slots = (size_t)c->cap_users << c->bdshift;
    const size_t ucap = batch_ucap(c);
    for (int i = 0; i < kBatchSlots; ++i) {
        BatchSlot& b = c->bslot[lane * kBatchSlots + i];
        PIE_HIP(c, hipMalloc(&b.direct, slots * sizeof(BktRec)));
        PIE_HIP(c, hipMalloc(&b.direct_hi, slots * 4));
        PIE_HIP(c, hipMalloc(&b.uoff, ((size_t)c->cap_users + 2) * 8));
        PIE_HIP(c, hipMalloc(&b.urows, ucap * 4));
        PIE_HIP(c, hipMalloc(&b.umlo, ucap * 4));
        PIE_HIP(c, hipMalloc(&b.umhi, ucap * 4));
    }
    for (int i = 0; i < 3; ++i) {
        char*& sp = c->bspan[lane * 3 + i];
        PIE_HIP(c, hipMalloc(&sp, batch_span_bytes(c)));
        PIE_HIP(c, hipMemsetAsync(sp, 0, batch_span_bytes(c), lane_stream_of(c, lane)));
    }
    c->lane_span_next[lane] = 0;
    c->lane_alloc[lane] = true;
    return PIE_OK;
}

// per-query list storage of a batch slot for at least n_q queries (16 / 32 / 64); growing it drops what it held, so it is
// sized before anything of the batch is written to it
int ensure_lists(pie_ctx* c, BatchSlot& b, int n_q)
{
    const int want = n_q <= 16 ? 16 : n_q <= 32 ? 32 : kBatchMax;
    if (b.lists_q >= want) return PIE_OK;
    PIE_HIP(c, hipStreamSynchronize(c->stream));
    dfree(b.counts_ord); dfree(b.offsets); dfree(b.out_idx);
    b.lists_q = 0;
    for (int q = 0; q < kBatchMax; ++q) b.list_ok[q] = false;
    const size_t us = (size_t)batch_users_stride(c), os = (size_t)batch_out_stride(c);
    PIE_HIP(c, hipMalloc(&b.counts_ord, (size_t)want * us * 4));
    PIE_HIP(c, hipMalloc(&b.offsets, (size_t)want * us * 8));
    PIE_HIP(c, hipMalloc(&b.out_idx, (size_t)want * (os > 0 ? os : 1) * 4));
    b.lists_q = want;
    return PIE_OK;
}

// batches finish in the order they were begun, whatever lane they ran on
BatchSlot* oldest_batch(pie_ctx* c)
{
    if (c->b_flight == 0) return nullptr;
    const int lane = c->fifo[c->fifo_head];
    return &c->bslot[lane * kBatchSlots + (c->lane_next[lane] + kBatchSlots - c->lane_flight[lane]) % kBatchSlots];
}

// bookkeeping of a batch that has been begun on `lane` in slot lane_next[lane]
void batch_begun(pie_ctx* c, int lane)
{
    c->fifo[(c->fifo_head + c->b_flight) % (kLaneMax * kBatchSlots)] = (unsigned char)lane;
    c->b_flight++;
    c->lane_flight[lane]++;
    c->lane_next[lane] = (c->lane_next[lane] + 1) % kBatchSlots;
}

// ... and of the oldest batch leaving the flight
void batch_left(pie_ctx* c, BatchSlot& b)
{
    b.in_flight = false;
    c->fifo_head = (c->fifo_head + 1) % (kLaneMax * kBatchSlots);
    c->b_flight--;
    c->lane_flight[b.lane]--;
    if (c->b_flight == 0) c->idle_epoch++;
}

// Work about to be queued on the context's MAIN stream reads what batch b's kernels wrote on a lane's stream.  The host has
// seen b's summary, but the launch that carried b's tail (the next table pass of that lane rides in it) may still be running,
// and plain stores sit in the XCDs' L2s until a kernel ends: the main stream waits for what the lane has queued so far.
int order_after_batch(pie_ctx* c, BatchSlot& b)
{
    if (b.main_ordered || b.stream == c->stream || !b.stream) { b.main_ordered = true; return PIE_OK; }
    if (!c->lane_event[b.lane][1]) PIE_HIP(c, hipEventCreateWithFlags(&c->lane_event[b.lane][1], hipEventDisableTiming));
    PIE_HIP(c, hipEventRecord(c->lane_event[b.lane][1], b.stream));
    PIE_HIP(c, hipStreamWaitEvent(c->stream, c->lane_event[b.lane][1], 0));
    b.main_ordered = true;
    return PIE_OK;
}

void fill_tail_args(pie_ctx* c, BatchSlot& b, UnionTailArgs& t)
{
    t.n_q = b.n_q;
    t.n_users = c->n_users;
    t.tiles = (c->n_users + kK1Threads - 1) / kK1Threads;
    t.dshift = b.dshift;
    t.span = b.span;
    t.tiles_off = (long long)span_counts_bytes(c);
    t.ctl_off = (long long)(span_counts_bytes(c) + span_tiles_bytes(c) + span_parts_bytes());
    t.summary_off = t.ctl_off + 128;
    t.mq_off = (long long)counts_span(c);
    t.zero_span = b.zero_span;   // the next-but-one batch's span
    t.zero_total16 = (long long)(batch_span_bytes(c) / 16);
    t.direct = b.direct;
    t.direct_hi = b.direct_hi;
    t.uoff = b.uoff; t.urows = b.urows; t.umlo = b.umlo; t.umhi = b.umhi;
    t.host = b.bh_dev;
    t.seq = b.seq;
    t.msg = b.msg_kind == 2 ? b.msg : nullptr;
    t.u_pad = b.msg_u_pad;
    t.msg_cap = b.msg_cap;
}

// the predicate tables of a batch (pie_kernels.h BatchTables) over the queries that take part in the pass
void fill_batch_tables(const pie_ctx* c, const BatchSlot& b, const pie_query* qs, const unsigned* nk, BatchTables& t)
{
    int by_now[kBatchMax], by_cut[kBatchMax], nb = 0;
    for (int q = 0; q < b.n_q; ++q)
        if (!b.fallback[q]) { by_now[nb] = q; by_cut[nb] = q; ++nb; }
    std::stable_sort(by_now, by_now + nb, [&](int x, int y) { return qs[x].now < qs[y].now; });
    std::stable_sort(by_cut, by_cut + nb, [&](int x, int y) { return qs[x].cutoff < qs[y].cutoff; });
    unsigned long long lv = 0, wn = 0;
    t.live[0] = 0;
    t.win[0] = 0;
    for (int i = 0; i < kBatchMax; ++i) {
        if (i < nb) {
            t.now[i] = qs[by_now[i]].now;
            t.nk[i] = nk[by_now[i]];
            t.cutoff[i] = qs[by_cut[i]].cutoff;
            lv |= 1ull << by_now[i];
            wn |= 1ull << by_cut[i];
        } else {
            t.now[i] = INT64_MAX;
            t.nk[i] = ~0u;
            t.cutoff[i] = INT64_MAX;
        }
        t.live[i + 1] = lv;
        t.win[i + 1] = wn;
    }
    for (int d = 0; d < 64; ++d) t.disc[d] = 0;
    const unsigned long long table = c->n_disc >= 64 ? ~0ull : ((1ull << c->n_disc) - 1ull);
    for (int i = 0; i < nb; ++i) { // every set bit of every query's mask, once
        const int q = by_now[i];
        for (unsigned long long m = qs[q].mask & table; m; m &= m - 1) t.disc[__builtin_ctzll(m)] |= 1ull << q;
    }
}

void launch_batch_k2(pie_ctx* c, BatchSlot& b, hipStream_t s)
{
    UnionTailArgs t;
    fill_tail_args(c, b, t);
    if (t.n_q > 32) hipLaunchKernelGGL(k_union_tail<true>, dim3((unsigned)t.tiles), dim3(kK1Threads), 0, s, t);
    else hipLaunchKernelGGL(k_union_tail<false>, dim3((unsigned)t.tiles), dim3(kK1Threads), 0, s, t);
    b.k2_pending = false;
}

int batch_begin(pie_ctx* c, const pie_query* qs, int n_q, int msg_kind, int* msg, long long msg_stride, int u_pad, long long msg_cap,
                int* msg_counts, long long msg_counts_stride)
{
    if (c->cap_rows == 0) return fail(c, PIE_E_STATE, "no table loaded");
    if (!qs || n_q < 1 || n_q > kBatchMax) return fail(c, PIE_E_INVAL, "a batch holds 1..%d queries (got %d)", kBatchMax, n_q);
    if (c->n_flight) return fail(c, PIE_E_STATE, "a single scan is in flight: finish it before beginning a batch");
    PIE_PROF_START();
    if (c->key_rebuild && c->b_flight == 0) {
        int rc = build_keys(c, 0, true);
        if (rc) return rc;
    }
    const bool ord_batch = batch_supported(c) && ordered_batch_wanted(c) && !c->ord_lists_only;
    bool unsupported = !ord_batch && (!batch_supported(c) || c->key_poor || c->batch_poor);
    // dense queries (the key histogram bounds their live rows above a tenth of the table) do not belong in a batch:
    // they would make every row a candidate for all queries; they run on the general path
    bool dense_q[kBatchMax];
    int n_batched = 0;
    bool fine = (c->k1_keyed & 0x800) && !c->fkey_poor;
    if (!unsupported) {
        const bool hist_ok = !c->key_dirty && c->key_hist_rows == c->n;
        for (int q = 0; q < n_q; ++q) {
            dense_q[q] = hist_ok && (double)rows_keyed_at_or_above(c, qs[q].now) >= kLiveFirstBelow * (double)c->n;
            if (!dense_q[q]) {
                ++n_batched;
                if (qs[q].now < c->fkey_base) fine = false;
            }
        }
        // nothing left to batch (every query is dense): no table pass at all — a pass whose smallest key is the impossible
        // one would still make nearly every row a candidate
        if (n_batched == 0) unsupported = true;
    }
    // the lane: batches are dealt round robin; a batch on the ordered run (its staging is the scan slots') stays on lane 0 (one
    // that only falls back queues nothing here: any lane's slot will do)
    int lane = 0;
    {
        const int lanes = ord_batch ? 1 : c->n_lanes;
        int tried = 0;
        lane = lanes > 1 ? c->lane_rr % lanes : 0;
        while (tried < lanes && c->lane_flight[lane] >= kBatchSlots) { lane = (lane + 1) % lanes; ++tried; }
        if (tried == lanes)
            return fail(c, PIE_E_STATE, "%d batches are already in flight (%d per lane, %d lane%s): call pie_scan_batch_finish first", c->b_flight,
                        kBatchSlots, lanes, lanes > 1 ? "s" : "");
        if (lanes > 1) c->lane_rr = (lane + 1) % lanes;
    }
    BatchSlot* const lane_slots = c->bslot + lane * kBatchSlots;
    BatchSlot& b = lane_slots[c->lane_next[lane]];
    hipStream_t s = lane_stream_of(c, lane);
    b.lane = lane;
    b.n_q = n_q;
    b.have_result = false;
    b.union_ok = b.union_part = false;
    b.mu = 0;
    if (c->bres == &b) c->bres = nullptr;
    b.msg_kind = msg ? msg_kind : 0;
    b.msg = msg; b.msg_stride = msg_stride; b.msg_u_pad = u_pad; b.msg_cap = msg_cap;
    b.msg_counts = msg_counts; b.msg_counts_stride = msg_counts_stride;
    for (int q = 0; q < n_q; ++q) { b.q[q] = qs[q]; b.fallback[q] = false; b.idx_of[q] = nullptr; b.list_ok[q] = false; }
    b.ev_index = -1;
    b.ordered = false;
    b.unsupported = unsupported;
    b.stream = s;
    b.main_ordered = lane == 0;
    if (!unsupported)
        for (int q = 0; q < n_q; ++q) b.fallback[q] = dense_q[q];
    PIE_PROF_TICK(0);
    if (b.unsupported) { // finish() runs every query on the general path
        for (int q = 0; q < n_q; ++q) b.fallback[q] = true;
        b.in_flight = true;
        b.k2_pending = false;
        b.main_ordered = true; // nothing of it runs on the lane's stream
        batch_begun(c, lane);
        return PIE_OK;
    }
    if (c->bdshift_want > c->bdshift && c->b_flight == 0 && any_lane_alloc(c)) {
        // larger union buckets for every user (a finished batch found one that outgrew its slots): both slots' arrays are
        // replaced; their contents are per-batch scratch
        int rc0 = sync_all(c);
        if (rc0) return rc0;
        const size_t slots = (size_t)c->cap_users << c->bdshift_want;
        if (slots * sizeof(BktRec) <= kDirectMaxBytes) {
            c->bdshift = c->bdshift_want;
            for (BatchSlot& x : c->bslot) {
                dfree(x.direct); dfree(x.direct_hi); dfree(x.uoff); dfree(x.urows); dfree(x.umlo); dfree(x.umhi);
                x.have_result = false;
            }
            for (char*& sp : c->bspan) dfree(sp);
            c->bres = nullptr;
            for (bool& a : c->lane_alloc) a = false;
        } else {
            c->bdshift_want = c->bdshift;
        }
    }
    int rc = ord_batch ? PIE_OK : ensure_batch(c, lane);
    if (rc) return rc;
    if (lane > 0 && c->lane_epoch[lane] != c->idle_epoch) { // see idle_epoch
        PIE_HIP(c, hipEventRecord(c->lane_event[lane][0], c->stream));
        PIE_HIP(c, hipStreamWaitEvent(s, c->lane_event[lane][0], 0));
        c->lane_epoch[lane] = c->idle_epoch;
    }
    b.fine_key = fine;
    b.dshift = c->bdshift;
    if (c->profiling && (c->scans_begun % (unsigned long long)c->profile_every) == 0) {
        if (c->ring_used == kEventRing) {
            rc = resolve_events(c);
            if (rc) return rc;
        }
        if (c->ring_used < kEventRing) {
            if ((int)c->ring.size() <= c->ring_used) {
                ScanEvents e{};
                PIE_HIP(c, hipEventCreate(&e.e0));
                PIE_HIP(c, hipEventCreate(&e.e1));
                PIE_HIP(c, hipEventCreate(&e.e2));
                c->ring.push_back(e);
            }
            b.ev_index = c->ring_used++;
        }
    }
    if (ord_batch) {
        // the table's batches do not fit the general pass (skewed users): ONE pass over the run's key column for the batch (pie_ordered.h, the batched union form)
        rc = ord_batch_alloc(c);
        if (rc) return rc;
        rc = ensure_batch(c, 0);
        if (rc) return rc;
        BatchSlot& prev = lane_slots[(c->lane_next[lane] + kBatchSlots - 1) % kBatchSlots];
        if (c->lane_flight[lane] >= 1 && prev.in_flight && prev.k2_pending) launch_batch_k2(c, prev, s); // nothing carries it along
        b.seq = ++c->bseq_counter;
        c->scans_begun++;
        if (b.ev_index >= 0) PIE_HIP(c, hipEventRecord(c->ring[b.ev_index].e0, s));
        launch_ordered_union(c, b, s, qs, fine);
        PIE_HIP(c, hipGetLastError());
        b.ordered = true;
        b.k2_pending = false;
        b.in_flight = true;
        batch_begun(c, lane);
        return PIE_OK;
    }
    // spans (the lane's own three): this batch's and the one its tail zeroes
    char** const lane_spans = c->bspan + lane * 3;
    b.span = lane_spans[c->lane_span_next[lane]];
    c->lane_span_next[lane] = (c->lane_span_next[lane] + 1) % 3;
    b.zero_span = lane_spans[(c->lane_span_next[lane] + 1) % 3];
    b.seq = ++c->bseq_counter;
    const int plan = fine ? 3 : 2;
    b.k1_blocks = c->plan_blocks[plan];
    c->scans_begun++;
    BatchSlot& other = lane_slots[(c->lane_next[lane] + kBatchSlots - 1) % kBatchSlots]; // the batch begun just before this one on this lane
    const bool tail_waits = c->lane_flight[lane] >= 1 && other.in_flight && other.k2_pending && !other.ordered;
    const bool ride = tail_waits && !c->no_ride;
    if (tail_waits && !ride) launch_batch_k2(c, other, s);
    if (b.ev_index >= 0) PIE_HIP(c, hipEventRecord(c->ring[b.ev_index].e0, s));
    PIE_PROF_TICK(1);
#define PIE_BATCH_RIDE(KT, HI)                                                                                          \
    hipLaunchKernelGGL((k_scan_batch_with_tail<8, true, KT, HI>), dim3((unsigned)(b.k1_blocks + t.tiles)), dim3(kK1Threads), 0, s, a, t)
#define PIE_BATCH(KT, KEYPTR, KEYFN, IMPOSSIBLE)                                                                         \
    do {                                                                                                                \
        BatchScanArgs<KT> a;                                                                                            \
        a.pay = c->d_pay; a.end = c->d_end; a.key = KEYPTR; a.n = c->n; a.rows_per_block = c->plan_rows[plan];          \
        a.n_users = c->n_users; a.n_q = n_q; a.dshift = c->bdshift;                                                     \
        a.counts = reinterpret_cast<int*>(b.span);                                                                      \
        a.summary = reinterpret_cast<Summary*>(b.span + span_counts_bytes(c) + span_tiles_bytes(c) + span_parts_bytes() + 128); \
        a.direct = b.direct; a.direct_hi = b.direct_hi; a.run_shift = c->run_shift;                                     \
        unsigned nk[kBatchMax];                                                                                         \
        unsigned mk = IMPOSSIBLE;                                                                                       \
        for (int q = 0; q < n_q; ++q) {                                                                                 \
            nk[q] = KEYFN(c, qs[q].now);                                                                                \
            if (!b.fallback[q] && nk[q] < mk) mk = nk[q];                                                               \
        }                                                                                                               \
        a.min_key = mk;                                                                                                 \
        fill_batch_tables(c, b, qs, nk, a.tab);                                                                         \
        PIE_PROF_TICK(2);                                                                                               \
        if (ride) {                                                                                                     \
            UnionTailArgs t;                                                                                            \
            fill_tail_args(c, other, t);                                                                                \
            if (t.n_q > 32) PIE_BATCH_RIDE(KT, true);                                                                   \
            else PIE_BATCH_RIDE(KT, false);                                                                             \
            other.k2_pending = false;                                                                                   \
        } else {                                                                                                        \
            hipLaunchKernelGGL((k_scan_batch<8, true, KT>), dim3((unsigned)b.k1_blocks), dim3(kK1Threads), 0, s, a);    \
        }                                                                                                               \
    } while (0)
    // a query that falls back is not in the tables: it selects nothing here
    if (fine) PIE_BATCH(fkey_t, c->d_fkey, host_fine_key_of, 0xFFu);
    else PIE_BATCH(lkey_t, c->d_key, host_key_of, 0xFFFFu);
#undef PIE_BATCH
#undef PIE_BATCH_RIDE
    PIE_PROF_TICK(3);
    PIE_HIP(c, hipGetLastError());
    if (b.ev_index >= 0) PIE_HIP(c, hipEventRecord(c->ring[b.ev_index].e1, s));
    b.k2_pending = true;
    b.in_flight = true;
    batch_begun(c, lane);
    PIE_PROF_TICK(4);
    return PIE_OK;
}

// where query q's row list of `m` rows lives in the list storage: its strided region, or a buffer of its own when it is longer
int batch_list_dst(pie_ctx* c, BatchSlot& b, int q, long long m, int** dst_out)
{
    int* dst = b.out_idx ? b.out_idx + (long long)q * batch_out_stride(c) : nullptr;
    if (!dst || m > batch_out_stride(c)) {
        if (b.over_cap[q] < m || !b.over_idx[q]) {
            PIE_HIP(c, hipStreamSynchronize(c->stream)); // an earlier copy may still be reading the old buffer's neighbours: keep it simple
            dfree(b.over_idx[q]);
            b.over_cap[q] = 0;
            PIE_HIP(c, hipMalloc(&b.over_idx[q], (size_t)(m > 0 ? m : 1) * 4));
            b.over_cap[q] = m > 0 ? m : 1;
        }
        dst = b.over_idx[q];
    }
    *dst_out = dst;
    return PIE_OK;
}

// per-query message [off[0..u_pad] | M | rows] of a query whose lists are in the list storage
void batch_pack_list_msg(pie_ctx* c, BatchSlot& b, int q)
{
    const long long m = (long long)b.last[q].m;
    const long long total = (long long)b.msg_u_pad + 2 + (m < b.msg_cap ? m : b.msg_cap);
    unsigned blocks = (unsigned)((total + 255) / 256);
    if (blocks > (unsigned)c->n_cus * 8) blocks = (unsigned)c->n_cus * 8;
    hipLaunchKernelGGL(k_pack_lists, dim3(blocks ? blocks : 1u), dim3(256), 0, c->stream, b.offsets + (long long)q * batch_users_stride(c), c->n_users,
                       b.msg_u_pad, m, b.idx_of[q], b.msg_cap, b.msg + (long long)q * b.msg_stride);
}

// Queries of a batch on the general path, two scans in flight (the offsets kernel of one rides in the next one's table
// pass); every result is copied into the batch's list storage in stream order, one wait at the end.
int batch_fallback_copy(pie_ctx* c, BatchSlot& b, int q)
{
    Slot& sl = *c->res;
    hipStream_t s = c->stream;
    const long long us = batch_users_stride(c);
    PIE_HIP(c, hipMemcpyAsync(b.counts_ord + (long long)q * us, sl.counts_ord, (size_t)c->n_users * 4, hipMemcpyDeviceToDevice, s));
    PIE_HIP(c, hipMemcpyAsync(b.offsets + (long long)q * us, sl.offsets, ((size_t)c->n_users + 1) * 8, hipMemcpyDeviceToDevice, s));
    const long long m = (long long)sl.last.m;
    int* dst = nullptr;
    int rc = batch_list_dst(c, b, q, m, &dst);
    if (rc) return rc;
    if (m) PIE_HIP(c, hipMemcpyAsync(dst, sl.out_idx, (size_t)m * 4, hipMemcpyDeviceToDevice, s));
    b.idx_of[q] = dst;
    b.list_ok[q] = true;
    if (b.msg_kind == 1) {
        rc = pie_pack_results_device(c, b.msg + (long long)q * b.msg_stride, (size_t)b.msg_u_pad, (size_t)b.msg_cap);
        if (rc) return rc;
    }
    if (b.msg_counts)
        PIE_HIP(c, hipMemcpyAsync(b.msg_counts + (long long)q * b.msg_counts_stride, sl.counts_ord, (size_t)c->n_users * 4, hipMemcpyDefault, s));
    b.last[q] = sl.last;
    return PIE_OK;
}

int batch_fallback_many(pie_ctx* c, BatchSlot& b, const int* list, int n_list)
{
    if (n_list == 0) return PIE_OK;
    const unsigned long long keep_mask = c->disc_mask;
    auto begin = [&](int q) {
        c->disc_mask = b.q[q].mask;
        return scan_begin(c, b.q[q].now, b.q[q].cutoff);
    };
    int rc = begin(list[0]);
    for (int i = 0; i < n_list && rc == PIE_OK; ++i) {
        if (i + 1 < n_list) rc = begin(list[i + 1]);
        if (rc == PIE_OK) rc = scan_finish(c);
        if (rc == PIE_OK) rc = batch_fallback_copy(c, b, list[i]);
    }
    c->disc_mask = keep_mask;
    if (rc != PIE_OK) { // leave no scan in flight behind an error
        while (c->n_flight) (void)scan_finish(c);
        return rc;
    }
    PIE_HIP(c, hipStreamSynchronize(c->stream));
    for (Slot& sl : c->slot) sl.have_result = false; // implementation details of the batch, not feed results of their own
    c->res = nullptr;
    return PIE_OK;
}

// counts / offsets / row lists of the listed queries out of the batch's union (k_mat_*); returns with the lists complete in
// stream order and, for sync = true, the per-query maxima read back
int batch_materialize(pie_ctx* c, BatchSlot& b, const int* list, int n_list)
{
    if (n_list == 0) return PIE_OK;
    if (!b.union_part) return fail(c, PIE_E_STATE, "the batch has no union to take query lists from");
    int rc = ensure_lists(c, b, b.n_q);
    if (rc) return rc;
    hipStream_t s = c->stream;
    const int tiles = (c->n_users + 255) / 256;
    if (c->mat_tiles < tiles || !c->d_mat_tile) {
        PIE_HIP(c, hipStreamSynchronize(s));
        dfree(c->d_mat_tile); dfree(c->d_mat_qmax);
        const int cap_tiles = (c->cap_users + 255) / 256 + 1;
        PIE_HIP(c, hipMalloc(&c->d_mat_tile, (size_t)kBatchMax * ((size_t)cap_tiles + 1) * 8));
        PIE_HIP(c, hipMalloc(&c->d_mat_qmax, (size_t)kBatchMax * 4));
        c->mat_tiles = cap_tiles;
    }
    MatArgs a;
    a.n_users = c->n_users; a.tiles = tiles; a.n_list = n_list;
    a.uoff = b.uoff; a.urows = b.urows; a.umlo = b.umlo; a.umhi = b.n_q > 32 ? b.umhi : nullptr;
    a.counts = b.counts_ord; a.offsets = b.offsets; a.users_stride = batch_users_stride(c);
    a.tile_sum = c->d_mat_tile; a.qmax = c->d_mat_qmax;
    for (int i = 0; i < n_list; ++i) {
        const int q = list[i];
        int* dst = nullptr;
        rc = batch_list_dst(c, b, q, (long long)b.last[q].m, &dst);
        if (rc) return rc;
        a.q_of[i] = (unsigned char)q;
        a.out[i] = dst;
        a.out_cap[i] = (long long)b.last[q].m;
        b.idx_of[q] = dst;
    }
    PIE_HIP(c, hipMemsetAsync(c->d_mat_qmax, 0, (size_t)kBatchMax * 4, s));
    hipLaunchKernelGGL(k_mat_count, dim3((unsigned)tiles, (unsigned)n_list), dim3(256), 0, s, a);
    hipLaunchKernelGGL(k_mat_prefix, dim3((unsigned)n_list), dim3(1024), 0, s, a);
    hipLaunchKernelGGL(k_mat_write, dim3((unsigned)tiles, (unsigned)n_list), dim3(256), 0, s, a);
    PIE_HIP(c, hipGetLastError());
    unsigned int qmax[kBatchMax];
    PIE_HIP(c, hipMemcpyAsync(qmax, c->d_mat_qmax, (size_t)n_list * 4, hipMemcpyDeviceToHost, s));
    PIE_HIP(c, hipStreamSynchronize(s));
    for (int i = 0; i < n_list; ++i) {
        b.last[list[i]].max_count = qmax[i];
        b.list_ok[list[i]] = true;
    }
    return PIE_OK;
}

// the lists of query qi of a finished batch, materialised if need be
int batch_need_list(pie_ctx* c, BatchSlot& b, int qi)
{
    if (b.list_ok[qi]) return PIE_OK;
    return batch_materialize(c, b, &qi, 1);
}

int batch_pack_union(pie_ctx* c, BatchSlot& b, void* dst_i32, size_t u_pad, size_t cap);

int batch_finish(pie_ctx* c, int* ready_out)
{
    if (ready_out) *ready_out = 0;
    BatchSlot* bp = oldest_batch(c);
    if (!bp) return fail(c, PIE_E_STATE, "pie_scan_batch_finish without pie_scan_batch_begin");
    BatchSlot& b = *bp;
    hipStream_t s = b.stream ? b.stream : c->stream; // the lane the batch runs on
    bool all_ready = true;
    PIE_PROF_START();
    if (!b.unsupported) {
        if (b.k2_pending) {
            launch_batch_k2(c, b, s);
            PIE_HIP(c, hipGetLastError());
        }
        PIE_PROF_TICK(5);
        // wait for the batch's summary (mapped host memory, seq last); bounded like the single-scan wait
        timespec t0{};
        clock_gettime(CLOCK_MONOTONIC, &t0);
        {
            volatile unsigned long long* seq = &b.bh->seq;
            unsigned long long spins = 0;
            while (*seq != b.seq) {
                __builtin_ia32_pause();
                if ((++spins & 0x3FFFF) == 0) {
                    hipError_t e = hipStreamQuery(s);
                    timespec t1{};
                    clock_gettime(CLOCK_MONOTONIC, &t1);
                    const double waited_ms = (double)(t1.tv_sec - t0.tv_sec) * 1e3 + (double)(t1.tv_nsec - t0.tv_nsec) * 1e-6;
                    const bool bad = e != hipSuccess && e != hipErrorNotReady;
                    if (bad || waited_ms > c->wait_deadline_ms || (e == hipSuccess && *seq != b.seq && waited_ms > 1000.0)) {
                        batch_left(c, b);
                        if (bad) return fail(c, PIE_E_HIP, "batched scan failed: %s", hipGetErrorString(e));
                        return fail(c, PIE_E_HIP, "batch summary not published within %.0f ms (PIE_WAIT_DEADLINE_MS): kernel hung?", waited_ms);
                    }
                }
            }
        }
        __atomic_thread_fence(__ATOMIC_ACQUIRE);
        PIE_PROF_TICK(6);
        bool overflow = false;
        {
            const Summary us = b.bh->s;
            b.mu = us.m;
            // a union bucket outgrew its slots (the general pass) / the union its arrays (the ordered run): rows were dropped, and
            // nothing says which queries they belonged to, so every query of the batch is rerun on the general path
            overflow = us.n_over > 0 || us.bad_rows > 0;
            for (int q = 0; q < b.n_q; ++q) {
                b.last[q] = us;                       // cand, chunk_max, bad_rows, n_over: the pass's; max_count: the union's
                b.last[q].m = b.bh->mq[q];
                if (overflow) b.fallback[q] = true;
            }
            if (us.n_over > 0 && b.ordered) c->ord_lists_only = true; // the union outgrew its arrays on the run: its batches run as single scans from now on
            else if (us.n_over > 0) {
                if (c->bdshift < kUnionShiftMax) c->bdshift_want = c->bdshift + 1;
                else c->batch_poor = true; // a user's rows do not fit 64 slots: this table's batches go straight to the general path
            }
            b.union_part = !overflow;
            b.union_ok = !overflow;
            for (int q = 0; q < b.n_q; ++q) b.union_ok = b.union_ok && !b.fallback[q];
            if (!b.ordered) choose_run_shift(c, us.cand, us.chunk_max, b.fine_key);
        }
        if (b.ev_index >= 0) PIE_HIP(c, hipEventRecord(c->ring[b.ev_index].e2, s));
    }
    batch_left(c, b);
    {
        int list[kBatchMax], n_list = 0;
        for (int q = 0; q < b.n_q; ++q)
            if (b.fallback[q]) list[n_list++] = q;
        if (n_list) {
            all_ready = false;
            int rc = order_after_batch(c, b); // what follows runs on the main stream
            if (rc) return rc;
            rc = ensure_lists(c, b, b.n_q);
            if (rc) return rc;
            rc = batch_fallback_many(c, b, list, n_list);
            if (rc) return rc;
        }
    }
    b.have_result = true;
    c->bres = &b;
    c->last_was_batch = true;
    for (int q = 0; q < b.n_q; ++q)
        if (b.last[q].bad_rows) return fail(c, PIE_E_INVAL, "query %d: %u selected rows carry a user id outside [0, %d)", q, b.last[q].bad_rows, c->n_users);
    if (b.msg_kind == 1 || b.msg_counts) {
        // per-query messages of a batch on the general pass: the lists are materialised from the union, then packed
        all_ready = false;
        int list[kBatchMax], n_list = 0;
        for (int q = 0; q < b.n_q; ++q)
            if (!b.list_ok[q]) list[n_list++] = q;
        int rc = order_after_batch(c, b);
        if (rc) return rc;
        rc = batch_materialize(c, b, list, n_list);
        if (rc) return rc;
        for (int i = 0; i < n_list; ++i) {
            const int q = list[i];
            if (b.msg_kind == 1) batch_pack_list_msg(c, b, q);
            if (b.msg_counts)
                PIE_HIP(c, hipMemcpyAsync(b.msg_counts + (long long)q * b.msg_counts_stride, b.counts_ord + (long long)q * batch_users_stride(c),
                                          (size_t)c->n_users * 4, hipMemcpyDefault, s));
        }
        PIE_HIP(c, hipGetLastError());
    }
    if (b.msg_kind == 2 && (!b.union_ok || b.ordered)) {
        // no tail wrote the message (the batch ran on the ordered run: copied from its union; or queries fell back: merged from the lists)
        all_ready = false;
        int rc = order_after_batch(c, b);
        if (rc) return rc;
        rc = batch_pack_union(c, b, b.msg, (size_t)b.msg_u_pad, (size_t)b.msg_cap);
        if (rc) return rc;
    }
    if (ready_out) *ready_out = b.msg_kind ? (all_ready ? 1 : 0) : 1;
    PIE_PROF_TICK(7);
    return PIE_OK;
}

// The union message of a finished batch into caller memory, enqueued on the context's stream.  A batch whose union the tail
// produced: one copy kernel.  Otherwise (queries fell back, the ordered run) the per-query lists are merged per user
// (k_union_collect): 32 query bits, 32 union rows per user — beyond either the message says Mu = -1 (use the lists).
int batch_pack_union(pie_ctx* c, BatchSlot& b, void* dst_i32, size_t u_pad, size_t cap)
{
    hipStream_t s = c->stream;
    if (b.union_ok) {
        size_t total = u_pad + 2 + (size_t)(b.mu < cap ? b.mu : cap);
        size_t grid = (total + 255) / 256;
        if (grid > (size_t)c->n_cus * 8) grid = (size_t)c->n_cus * 8;
        hipLaunchKernelGGL(k_union_pack, dim3((unsigned)(grid ? grid : 1)), dim3(256), 0, s, c->n_users, (int)u_pad, b.uoff, b.urows, b.umlo,
                           b.n_q > 32 ? b.umhi : (const unsigned*)nullptr, (long long)cap, (int*)dst_i32);
        PIE_HIP(c, hipGetLastError());
        return PIE_OK;
    }
    {   // every query's list is needed: the ones still only in the (partial) union are materialised
        int list[kBatchMax], n_list = 0;
        for (int q = 0; q < b.n_q; ++q)
            if (!b.list_ok[q]) list[n_list++] = q;
        int rc = batch_materialize(c, b, list, n_list);
        if (rc) return rc;
    }
    const size_t padded = (((size_t)c->cap_users + 1023) / 1024 + 1) * 1024, groups = padded / 1024 + 2;
    if (c->union_users < c->cap_users || !c->d_union) {
        if ((size_t)c->cap_users * kUnionMax * sizeof(UnionRow) > ((size_t)4 << 30)) return fail(c, PIE_E_NOMEM, "union scratch for %d users exceeds 4 GiB", c->cap_users);
        PIE_HIP(c, hipStreamSynchronize(s));
        dfree(c->d_union); dfree(c->d_union_cnt); dfree(c->d_union_local); dfree(c->d_union_off);
        PIE_HIP(c, hipMalloc(&c->d_union, (size_t)c->cap_users * kUnionMax * sizeof(UnionRow)));
        PIE_HIP(c, hipMalloc(&c->d_union_cnt, padded * 4));
        PIE_HIP(c, hipMalloc(&c->d_union_local, padded * 4));
        PIE_HIP(c, hipMalloc(&c->d_union_off, 2 * groups * 8 + 256 + 64));
        PIE_HIP(c, hipMemsetAsync(c->d_union_cnt, 0, padded * 4, s)); // entries behind the users stay zero for good
        PIE_HIP(c, hipMemsetAsync(c->d_union_off, 0, 2 * groups * 8 + 256 + 64, s)); // incl. the prefix kernel's block counter
        c->union_users = c->cap_users;
    }
    long long* gsum = c->d_union_off;
    long long* gbase = c->d_union_off + groups;
    Summary* usum = reinterpret_cast<Summary*>(reinterpret_cast<char*>(c->d_union_off + 2 * groups));
    int* over = reinterpret_cast<int*>(reinterpret_cast<char*>(usum) + 256);
    OrdCtl* uctl = reinterpret_cast<OrdCtl*>(reinterpret_cast<char*>(usum) + 256 + 16); // its counter returns to 0 by itself
    if (b.n_q > 32) { // the merged form carries 32 query bits
        const int one = 1;
        PIE_HIP(c, hipMemcpyAsync(over, &one, 4, hipMemcpyHostToDevice, s));
        PIE_HIP(c, hipStreamSynchronize(s));
    } else {
        PIE_HIP(c, hipMemsetAsync(over, 0, 4, s));
    }
    UnionLists lists{};
    for (int q = 0; q < b.n_q && q < 32; ++q) lists.idx[q] = b.idx_of[q];
    const unsigned ublocks = (unsigned)((c->n_users + kUnionThreads - 1) / kUnionThreads);
    hipLaunchKernelGGL(k_union_collect, dim3(ublocks ? ublocks : 1u), dim3(kUnionThreads), 0, s, b.n_q < 32 ? b.n_q : 32, c->n_users, b.offsets, batch_users_stride(c),
                       lists, c->d_start, c->d_union, c->d_union_cnt, over);
    long long n_groups = ((long long)c->n_users + 1023) >> 10;
    if (n_groups < 1) n_groups = 1;
    const unsigned pre_grid = (unsigned)(n_groups < (long long)c->n_cus * 4 ? n_groups : (long long)c->n_cus * 4);
    // a hundred groups at most: the one-launch form (the block that finishes last scans the group sums)
    hipLaunchKernelGGL(k_ord_prefix, dim3(pre_grid), dim3(256), 0, s, c->d_union_cnt, (long long)c->n_users, c->d_union_local, gsum, gbase,
                       uctl, usum, 0LL, 0LL, 0LL);
    unsigned wblocks = (unsigned)((u_pad + 2 + 255) / 256);
    if (wblocks > (unsigned)c->n_cus * 8) wblocks = (unsigned)c->n_cus * 8;
    hipLaunchKernelGGL(k_union_write, dim3(wblocks), dim3(256), 0, s, c->n_users, (int)u_pad, c->d_union_local, gbase, c->d_union_cnt, c->d_union, over,
                       (long long)cap, (int*)dst_i32);
    PIE_HIP(c, hipGetLastError());
    return PIE_OK;
}

} // namespace

extern "C" {

int pie_abi_version(void) { return PIE_ABI_VERSION; }

int pie_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

const char* pie_last_error(const pie_ctx* ctx) { return ctx ? ctx->err : g_create_error; }

int pie_ctx_create(int device_id, pie_ctx** ctx_out)
{
    if (!ctx_out) return fail(nullptr, PIE_E_INVAL, "ctx_out is NULL");
    *ctx_out = nullptr;
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n == 0)
        return fail(nullptr, PIE_E_NODEVICE, "no HIP device (%s); this library has no CPU path",
                    e == hipSuccess ? "device count 0" : hipGetErrorString(e));
    if (device_id < 0 || device_id >= n) return fail(nullptr, PIE_E_INVAL, "device %d outside [0, %d)", device_id, n);
    if ((e = hipSetDevice(device_id)) != hipSuccess)
        return fail(nullptr, PIE_E_NODEVICE, "hipSetDevice(%d): %s", device_id, hipGetErrorString(e));
    pie_ctx* c = new (std::nothrow) pie_ctx();
    if (!c) return fail(nullptr, PIE_E_NOMEM, "out of host memory");
    c->device = device_id;
    hipDeviceProp_t prop;
    if ((e = hipGetDeviceProperties(&prop, device_id)) != hipSuccess) {
        delete c;
        return fail(nullptr, PIE_E_NODEVICE, "hipGetDeviceProperties: %s", hipGetErrorString(e));
    }
    c->n_cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    bool ok = (e = hipStreamCreateWithFlags(&c->own_stream, hipStreamNonBlocking)) == hipSuccess;
    // the lanes' streams right behind the main one: the runtime deals its (four) hardware queues to streams in the order they
    // first appear, and a lane that ends up sharing a queue with another lane runs behind it, not beside it
    for (int l = 1; l < kLaneMax && ok; ++l) ok = (e = hipStreamCreateWithFlags(&c->lane_stream[l], hipStreamNonBlocking)) == hipSuccess;
    ok = ok &&
              (e = hipMalloc(&c->d_summary, sizeof(Summary))) == hipSuccess &&
              (e = hipMalloc(&c->d_range, 16)) == hipSuccess &&
              (e = hipMalloc(&c->d_hist, kKeyHistBins * sizeof(unsigned int))) == hipSuccess &&
              (e = hipHostMalloc(&c->h_summary, sizeof(Summary), hipHostMallocDefault)) == hipSuccess;
    for (Slot& s : c->slot) {
        ok = ok && (e = hipHostMalloc(&s.h_sum, sizeof(HostSummary), hipHostMallocMapped)) == hipSuccess &&
             (e = hipHostGetDevicePointer((void**)&s.h_sum_dev, s.h_sum, 0)) == hipSuccess;
        if (ok) memset(s.h_sum, 0, sizeof(HostSummary));
    }
    ok = ok && (e = hipHostMalloc(&c->ord.h_stale, 64, hipHostMallocMapped)) == hipSuccess &&
         (e = hipHostGetDevicePointer((void**)&c->ord.stale, c->ord.h_stale, 0)) == hipSuccess;
    if (ok) memset(c->ord.h_stale, 0, 64);
    for (BatchSlot& b : c->bslot) {
        ok = ok && (e = hipHostMalloc(&b.bh, sizeof(BatchHost), hipHostMallocMapped)) == hipSuccess &&
             (e = hipHostGetDevicePointer((void**)&b.bh_dev, b.bh, 0)) == hipSuccess;
        if (ok) memset(b.bh, 0, sizeof(BatchHost));
    }
    if (!ok) {
        fail(nullptr, PIE_E_NODEVICE, "context setup: %s", hipGetErrorString(e));
        delete c; // the few handles created so far go with the process: it has no usable GPU anyway
        return PIE_E_NODEVICE;
    }
    c->stream = c->own_stream;
    if (const char* v = getenv("PIE_K1_VARIANT")) { c->k1_variant = (int)strtol(v, nullptr, 0); c->k1_pinned = true; }
    if (const char* v = getenv("PIE_K1_LIVE_FIRST")) c->k1_live_first = (int)strtol(v, nullptr, 0);
    if (const char* v = getenv("PIE_FAST_PATH")) c->fast_env = atoi(v) != 0;
    if (const char* v = getenv("PIE_FUSED_ORDER")) c->no_fused_order = atoi(v) == 0;
    if (const char* v = getenv("PIE_K2_RIDE")) c->no_ride = atoi(v) == 0;
    if (const char* v = getenv("PIE_EXPIRED_COPY_TOTAL")) c->expired_copy_total = atoi(v) != 0;
    if (const char* v = getenv("PIE_ASYNC_MUTATIONS")) c->async_mutations = atoi(v) != 0;
    if (const char* v = getenv("PIE_BATCH_LANES")) { const int l = atoi(v); if (l >= 0 && l <= kLaneMax) c->lanes_want = l; }
    if (const char* v = getenv("PIE_ORDER_BLOCK")) { const int b = atoi(v); if (b == 256 || b == 512 || b == 1024) c->order_block = b; }
    if (const char* v = getenv("PIE_RUN_SHIFT")) { const int r = atoi(v); if (r >= 0 && r <= 3) { c->run_shift = r; c->run_shift_pinned = true; } }
    if (const char* v = getenv("PIE_ORDERED")) { const int m = atoi(v); if (m >= 0 && m <= 2) c->ord.mode = m; }
    if (const char* v = getenv("PIE_ORD_GRID")) { const int g = atoi(v); if (g >= 1 && g <= 64) c->ord.grid_mult = g; }
    if (const char* v = getenv("PIE_WAIT_DEADLINE_MS")) { const double d = atof(v); if (d > 0) c->wait_deadline_ms = d; }
    if (const char* v = getenv("PIE_K1_KEYED")) {
        const int k = (int)strtol(v, nullptr, 0);
        c->keyed_enabled = k != 0;
        if (k & 0x400) c->k1_keyed = k;
    }
    *ctx_out = c;
    return PIE_OK;
}

int pie_ctx_destroy(pie_ctx* c)
{
    if (!c) return PIE_OK;
#ifdef PIE_HOST_PROF
    g_prof.report();
#endif
    (void)hipSetDevice(c->device);
    (void)hipStreamSynchronize(c->stream);
    free_table(c);
    dfree(c->d_shard_rows);
    dfree(c->d_shard_users);
    dfree(c->d_stage);
    if (c->h_stage) (void)hipHostFree(c->h_stage);
    for (auto& a : c->astage) {
        if (a.h) (void)hipHostFree(a.h);
        if (a.d) (void)hipFree(a.d);
        if (a.ev) (void)hipEventDestroy(a.ev);
    }
    for (auto& e : c->ring) {
        (void)hipEventDestroy(e.e0); (void)hipEventDestroy(e.e1); (void)hipEventDestroy(e.e2);
    }
    for (Slot& s : c->slot) {
        if (s.h_sum) (void)hipHostFree(s.h_sum);
    }
    for (BatchSlot& b : c->bslot) {
        if (b.bh) (void)hipHostFree(b.bh);
    }
    if (c->ord.h_stale) (void)hipHostFree(c->ord.h_stale);
    if (c->d_summary) (void)hipFree(c->d_summary);
    if (c->d_range) (void)hipFree(c->d_range);
    if (c->d_hist) (void)hipFree(c->d_hist);
    if (c->h_summary) (void)hipHostFree(c->h_summary);
    if (c->own_stream) (void)hipStreamDestroy(c->own_stream);
    for (int l = 1; l < kLaneMax; ++l)
        if (c->lane_stream[l]) (void)hipStreamDestroy(c->lane_stream[l]);
    for (auto& pair : c->lane_event)
        for (hipEvent_t e : pair)
            if (e) (void)hipEventDestroy(e);
    delete c;
    return PIE_OK;
}

// Lanes of the batched scan: 1..4 pins the number, 0 chooses by the size of the resident table (whenever it changes).  May be
// called at any time: a batch in flight stays on the lane it was begun on.
int pie_set_batch_lanes(pie_ctx* c, int n_lanes)
{
    if (!c) return PIE_E_INVAL;
    if (n_lanes < 0 || n_lanes > kLaneMax) return fail(c, PIE_E_INVAL, "lanes outside 0..%d", kLaneMax);
    c->lanes_want = n_lanes;
    choose_lanes(c);
    return PIE_OK;
}

int pie_batch_lanes(pie_ctx* c) { return c ? c->n_lanes : 0; }

// how many more batches pie_scan_batch_begin would take right now
int pie_batch_room(pie_ctx* c)
{
    if (!c || c->n_flight) return 0;
    const bool ord_batch = batch_supported(c) && ordered_batch_wanted(c) && !c->ord_lists_only;
    const int lanes = ord_batch ? 1 : c->n_lanes;
    int room = 0;
    for (int l = 0; l < lanes; ++l) room += kBatchSlots - c->lane_flight[l];
    return room;
}

int pie_ctx_set_stream(pie_ctx* c, void* hip_stream)
{
    if (!c) return PIE_E_INVAL;
    if (c->n_flight || c->b_flight) return fail(c, PIE_E_STATE, "a scan is in flight");
    int rc = sync_all(c);
    if (rc) return rc;
    c->stream = hip_stream ? (hipStream_t)hip_stream : c->own_stream;
    return PIE_OK;
}

int pie_ctx_aux_stream(pie_ctx* c, void** hip_stream_out)
{
    if (!c || !hip_stream_out) return PIE_E_INVAL;
    *hip_stream_out = (void*)c->stream;
    return PIE_OK;
}

int pie_load_columns(pie_ctx* c, const int64_t* start, const int64_t* end, const int32_t* user, const int32_t* disc,
                     size_t n, int32_t n_users)
{
    if (!c) return PIE_E_INVAL;
    if (n > 0 && (!start || !end || !user || !disc)) return fail(c, PIE_E_INVAL, "NULL column pointer");
    PIE_HIP(c, hipSetDevice(c->device));
    int rc = ensure_capacity(c, (long long)n, n_users);
    if (rc) return rc;
    if (n > 0) {
        PIE_HIP(c, hipMemcpyAsync(c->d_start, start, n * 8, hipMemcpyHostToDevice, c->stream));
        PIE_HIP(c, hipMemcpyAsync(c->d_end, end, n * 8, hipMemcpyHostToDevice, c->stream));
        PIE_HIP(c, hipMemcpyAsync(c->d_user, user, n * 4, hipMemcpyHostToDevice, c->stream));
        PIE_HIP(c, hipMemcpyAsync(c->d_disc, disc, n * 4, hipMemcpyHostToDevice, c->stream));
    }
    rc = validate_users(c);
    if (rc) { c->n = 0; plan_k1(c); return rc; }
    return build_keys(c, 0);
}

int pie_append_rows(pie_ctx* c, const int64_t* start, const int64_t* end, const int32_t* user, const int32_t* disc,
                    size_t k, int32_t n_users)
{
    if (!c) return PIE_E_INVAL;
    if (k > 0 && (!start || !end || !user || !disc)) return fail(c, PIE_E_INVAL, "NULL column pointer");
    if (n_users < c->n_users) return fail(c, PIE_E_INVAL, "n_users may only grow (%d < %d)", n_users, c->n_users);
    PIE_HIP(c, hipSetDevice(c->device));
    const long long old_n = c->n;
    if (k > 0 && old_n > 0 && old_n + (long long)k <= c->cap_rows && n_users <= c->cap_users && c->key_ok && k <= ((size_t)1 << 24)) {
        // In-place path (room for the rows and the users, key columns in step): ONE staged upload, ONE kernel that writes the
        // four columns, both keys and the payload record of every new row and validates the user ids, ONE wait.  A login
        // burst costs tens of microseconds, not the half-dozen blocking calls of the general path below.
        if (c->n_flight || c->b_flight) return fail(c, PIE_E_STATE, "table change while a scan is in flight");
        if (old_n + (long long)k >= (1LL << 31) - 1) return fail(c, PIE_E_INVAL, "row count outside [0, 2^31 - 1)");
        if (c->async_mutations && !c->ord.valid) {
            // nothing to read back: user ids are checked here, the rows are staged and queued, the call returns (see AsyncStage)
            unsigned bad = 0;
            for (size_t i = 0; i < k; ++i) bad += (unsigned)user[i] >= (unsigned)n_users ? 1u : 0u;
            if (bad) return fail(c, PIE_E_INVAL, "%u rows carry a user id outside [0, %d)", bad, n_users);
            pie_ctx::AsyncStage* a = nullptr;
            int rca = async_stage(c, k * 24 + 64, &a);
            if (rca) return rca;
            memcpy(a->h, start, k * 8);
            memcpy(a->h + k * 8, end, k * 8);
            memcpy(a->h + k * 16, user, k * 4);
            memcpy(a->h + k * 20, disc, k * 4);
            hipStream_t s = c->stream;
            PIE_HIP(c, hipMemcpyAsync(a->d, a->h, k * 24, hipMemcpyHostToDevice, s));
            const unsigned grid = (unsigned)((k + 255) / 256) < (unsigned)c->n_cus * 8 ? (unsigned)((k + 255) / 256) : (unsigned)c->n_cus * 8;
            hipLaunchKernelGGL(k_append_rows, dim3(grid), dim3(256), 0, s, reinterpret_cast<const long long*>(a->d),
                               reinterpret_cast<const long long*>(a->d + k * 8), reinterpret_cast<const int*>(a->d + k * 16),
                               reinterpret_cast<const int*>(a->d + k * 20), (long long)k, old_n, n_users, c->d_start, c->d_end, c->d_user,
                               c->d_disc, c->d_key, c->key_base, c->key_shift, c->d_fkey, c->fkey_base, c->fkey_shift, c->d_pay,
                               &c->d_summary->bad_rows);
            PIE_HIP(c, hipGetLastError());
            PIE_HIP(c, hipEventRecord(a->ev, s));
            a->pending = true;
            c->n = old_n + (long long)k;
            ord_invalidate(c);
            if (n_users > c->n_users) set_user_count(c, n_users);
            c->key_dirty = true;
            c->res = nullptr;
            for (Slot& sl : c->slot) sl.have_result = false;
            c->bres = nullptr;
            plan_k1(c);
            return ensure_sel(c);
        }
        int rc0 = ensure_stage(c, k * 24 + 64);
        if (rc0) return rc0;
        char* h = c->h_stage;
        memcpy(h, start, k * 8);
        memcpy(h + k * 8, end, k * 8);
        memcpy(h + k * 16, user, k * 4);
        memcpy(h + k * 20, disc, k * 4);
        hipStream_t s = c->stream;
        PIE_HIP(c, hipMemcpyAsync(c->d_stage, h, k * 24, hipMemcpyHostToDevice, s));
        PIE_HIP(c, hipMemsetAsync(&c->d_summary->bad_rows, 0, sizeof(unsigned int), s));
        const unsigned grid = (unsigned)((k + 255) / 256) < (unsigned)c->n_cus * 8 ? (unsigned)((k + 255) / 256) : (unsigned)c->n_cus * 8;
        hipLaunchKernelGGL(k_append_rows, dim3(grid), dim3(256), 0, s, reinterpret_cast<const long long*>(c->d_stage),
                           reinterpret_cast<const long long*>(c->d_stage + k * 8), reinterpret_cast<const int*>(c->d_stage + k * 16),
                           reinterpret_cast<const int*>(c->d_stage + k * 20), (long long)k, old_n, n_users, c->d_start, c->d_end, c->d_user,
                           c->d_disc, c->d_key, c->key_base, c->key_shift, c->d_fkey, c->fkey_base, c->fkey_shift, c->d_pay,
                           &c->d_summary->bad_rows);
        // the ordered run takes rows that arrive in time order into the spare slots of their users' segments
        bool ord_kept = false;
        if (c->ord.valid && c->ord.rows == old_n && k <= (size_t)kOrdAppendMax && n_users <= c->ord.users && c->key_ok) {
            launch_ord_append(c, s, k, old_n, n_users, 1); // pend[] is all zero between appends (build and re-spread leave it so)
            ord_kept = true;
        }
        PIE_HIP(c, hipGetLastError());
        PIE_HIP(c, hipMemcpyAsync(c->h_summary, c->d_summary, sizeof(Summary), hipMemcpyDeviceToHost, s));
        PIE_HIP(c, hipStreamSynchronize(s));
        if (c->h_summary->bad_rows) { // the rows were written beyond n: the table itself is unchanged
            if (ord_kept) { // ... but some of them may sit in the run's spare slots
                ord_invalidate(c);
                c->ord.h_stale[0] = c->ord.h_stale[1] = 0;
            }
            return fail(c, PIE_E_INVAL, "%u rows carry a user id outside [0, %d)", c->h_summary->bad_rows, n_users);
        }
        c->n = old_n + (long long)k;
        if (ord_kept) {
            volatile unsigned int* st = c->ord.h_stale; // [0] rows out of time order, [1] rows whose segment was full
            if (st[0] == 0 && st[1] != 0) {
                int rc2 = ord_respread(c, k, old_n, n_users);
                if (rc2) return rc2;
            } else if (st[0] != 0) ord_invalidate(c, false, true);
            c->ord.h_stale[0] = c->ord.h_stale[1] = 0;
            if (c->ord.valid) {
                c->ord.rows = c->n;
                c->ord.held += (long long)k;
            }
        } else ord_invalidate(c);
        if (n_users > c->n_users) set_user_count(c, n_users);
        c->key_dirty = true;
        c->res = nullptr;
        for (Slot& sl : c->slot) sl.have_result = false;
        c->bres = nullptr;
        plan_k1(c);
        return ensure_sel(c);
    }
    int rc = ensure_capacity(c, old_n + (long long)k, n_users, old_n > 0 ? old_n : 1);
    if (rc) return rc;
    if (k > 0) {
        PIE_HIP(c, hipMemcpyAsync(c->d_start + old_n, start, k * 8, hipMemcpyHostToDevice, c->stream));
        PIE_HIP(c, hipMemcpyAsync(c->d_end + old_n, end, k * 8, hipMemcpyHostToDevice, c->stream));
        PIE_HIP(c, hipMemcpyAsync(c->d_user + old_n, user, k * 4, hipMemcpyHostToDevice, c->stream));
        PIE_HIP(c, hipMemcpyAsync(c->d_disc + old_n, disc, k * 4, hipMemcpyHostToDevice, c->stream));
    }
    rc = validate_users(c, old_n);
    if (rc) { c->n = old_n; plan_k1(c); return rc; }
    return build_keys(c, old_n);
}

int pie_gen_synthetic(pie_ctx* c, uint64_t seed, int64_t n_total, int64_t row0, int64_t n, int32_t n_users,
                      int32_t n_disc, uint32_t flags)
{
    if (!c) return PIE_E_INVAL;
    if (n < 0 || row0 < 0 || n_total < row0 + n || n_disc < 1 || n_disc > 64)
        return fail(c, PIE_E_INVAL, "bad synthetic-corpus shape");
    PIE_HIP(c, hipSetDevice(c->device));
    int rc = ensure_capacity(c, n, n_users);
    if (rc) return rc;
    if (n > 0) {
        hipLaunchKernelGGL(k_gen, dim3(c->n_cus * 8), dim3(256), 0, c->stream, seed, (long long)n_total, (long long)row0,
                           (long long)n, n_users, n_disc, flags, c->d_start, c->d_end, c->d_user, c->d_disc);
        PIE_HIP(c, hipGetLastError());
    }
    rc = build_keys(c, 0);
    if (rc) return rc;
    PIE_HIP(c, hipStreamSynchronize(c->stream));
    return PIE_OK;
}

int pie_gen_synthetic_cdf(pie_ctx* c, uint64_t seed, int64_t n_total, int64_t row0, int64_t n, int32_t n_users,
                          int32_t n_disc, uint32_t flags, const uint64_t* cdf)
{
    if (!c) return PIE_E_INVAL;
    if (!cdf) return fail(c, PIE_E_INVAL, "cdf is NULL");
    int rc = pie_gen_synthetic(c, seed, n_total, row0, n, n_users, n_disc, flags & ~2u);
    if (rc || n == 0) return rc;
    unsigned long long* d_cdf = nullptr;
    PIE_HIP(c, hipMalloc(&d_cdf, (size_t)n_users * 8));
    hipError_t e = hipMemcpyAsync(d_cdf, cdf, (size_t)n_users * 8, hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess) {
        hipLaunchKernelGGL(k_gen_users_cdf, dim3(c->n_cus * 8), dim3(256), 0, c->stream, seed, (long long)row0, (long long)n,
                           n_users, d_cdf, c->d_user);
        e = hipGetLastError();
    }
    hipError_t e2 = hipStreamSynchronize(c->stream);
    (void)hipFree(d_cdf);
    if (e != hipSuccess || e2 != hipSuccess)
        return fail(c, PIE_E_HIP, "pie_gen_synthetic_cdf: %s", hipGetErrorString(e != hipSuccess ? e : e2));
    c->key_ok = false; // the user column was rewritten: the payload column has to follow
    rc = build_keys(c, 0);
    if (rc) return rc;
    PIE_HIP(c, hipStreamSynchronize(c->stream));
    return PIE_OK;
}

// ---- on-disk column format: <dir>/start.i64, end.i64, user.i32, disc.i32 (raw little-endian arrays) and
// <dir>/header.json {"format":"pie-columns","version":1,"rows":N,"users":U}.  Contrast: the reference exports its whole
// SQLite image after every mutation (/root/reference/server/storage/sqlProvider.js:737-744) and keeps sessions in memory
// only (/root/reference/server/sessionStore.js:6).
namespace {
int write_file(pie_ctx* c, const std::string& path, const void* data, size_t bytes)
{
    const int fd = open(path.c_str(), O_WRONLY | O_CREAT | O_TRUNC, 0644);
    if (fd < 0) return fail(c, PIE_E_INVAL, "cannot create %s", path.c_str());
    const char* p = static_cast<const char*>(data);
    size_t left = bytes;
    while (left) {
        const ssize_t w = write(fd, p, left > (1u << 30) ? (1u << 30) : left);
        if (w <= 0) { close(fd); return fail(c, PIE_E_INVAL, "short write to %s", path.c_str()); }
        p += w;
        left -= (size_t)w;
    }
    close(fd);
    return PIE_OK;
}
} // namespace

int pie_save_columns(pie_ctx* c, const char* dir)
{
    if (!c || !dir) return PIE_E_INVAL;
    if (c->cap_rows == 0) return fail(c, PIE_E_STATE, "no table loaded");
    if (c->n_flight || c->b_flight) return fail(c, PIE_E_STATE, "a scan is in flight");
    PIE_HIP(c, hipSetDevice(c->device));
    (void)mkdir(dir, 0755);
    const size_t n = (size_t)c->n;
    std::vector<char> host(n * 8 + 8);
    const std::string base(dir);
    struct { const char* name; const void* dev; size_t width; } cols[4] = {
        {"/start.i64", c->d_start, 8}, {"/end.i64", c->d_end, 8}, {"/user.i32", c->d_user, 4}, {"/disc.i32", c->d_disc, 4}};
    for (auto& col : cols) {
        if (n) PIE_HIP(c, hipMemcpy(host.data(), col.dev, n * col.width, hipMemcpyDeviceToHost));
        int rc = write_file(c, base + col.name, host.data(), n * col.width);
        if (rc) return rc;
    }
    char hdr[256];
    const int len = snprintf(hdr, sizeof hdr, "{\"format\":\"pie-columns\",\"version\":1,\"rows\":%lld,\"users\":%d}\n", c->n, c->n_users);
    return write_file(c, base + "/header.json", hdr, (size_t)len);
}

int pie_load_columns_dir(pie_ctx* c, const char* dir)
{
    if (!c || !dir) return PIE_E_INVAL;
    const std::string base(dir);
    FILE* f = fopen((base + "/header.json").c_str(), "r");
    if (!f) return fail(c, PIE_E_INVAL, "no header.json under %s", dir);
    char hdr[512] = {0};
    const size_t got = fread(hdr, 1, sizeof hdr - 1, f);
    fclose(f);
    (void)got;
    long long rows = -1;
    int users = -1, version = -1;
    const char* pr = strstr(hdr, "\"rows\":");
    const char* pu = strstr(hdr, "\"users\":");
    const char* pv = strstr(hdr, "\"version\":");
    if (!strstr(hdr, "\"pie-columns\"") || !pr || !pu || !pv) return fail(c, PIE_E_INVAL, "%s/header.json is not a pie-columns header", dir);
    rows = atoll(pr + 7);
    users = atoi(pu + 8);
    version = atoi(pv + 10);
    if (version != 1 || rows < 0 || users < 1) return fail(c, PIE_E_INVAL, "unsupported header in %s (version %d)", dir, version);
    const size_t n = (size_t)rows;
    const char* names[4] = {"/start.i64", "/end.i64", "/user.i32", "/disc.i32"};
    const size_t width[4] = {8, 8, 4, 4};
    void* maps[4] = {nullptr, nullptr, nullptr, nullptr};
    int rc = PIE_OK;
    for (int k = 0; k < 4 && rc == PIE_OK; ++k) {
        const int fd = open((base + names[k]).c_str(), O_RDONLY);
        struct stat st;
        if (fd < 0 || fstat(fd, &st) != 0 || (size_t)st.st_size != n * width[k]) {
            if (fd >= 0) close(fd);
            rc = fail(c, PIE_E_INVAL, "%s%s missing or not %zu bytes", dir, names[k], n * width[k]);
            break;
        }
        if (n) {
            maps[k] = mmap(nullptr, n * width[k], PROT_READ, MAP_PRIVATE, fd, 0);
            if (maps[k] == MAP_FAILED) { maps[k] = nullptr; rc = fail(c, PIE_E_NOMEM, "mmap of %s%s failed", dir, names[k]); }
        }
        close(fd);
    }
    if (rc == PIE_OK)
        rc = pie_load_columns(c, (const int64_t*)maps[0], (const int64_t*)maps[1], (const int32_t*)maps[2], (const int32_t*)maps[3], n, users);
    for (int k = 0; k < 4; ++k)
        if (maps[k]) munmap(maps[k], n * width[k]);
    return rc;
}

int pie_read_columns(pie_ctx* c, int64_t* start, int64_t* end, int32_t* user, int32_t* disc, size_t n)
{
    if (!c) return PIE_E_INVAL;
    if ((long long)n > c->n) return fail(c, PIE_E_INVAL, "asked for %zu rows, table has %lld", n, c->n);
    PIE_HIP(c, hipSetDevice(c->device));
    if (n) {
        if (start) PIE_HIP(c, hipMemcpyAsync(start, c->d_start, n * 8, hipMemcpyDeviceToHost, c->stream));
        if (end) PIE_HIP(c, hipMemcpyAsync(end, c->d_end, n * 8, hipMemcpyDeviceToHost, c->stream));
        if (user) PIE_HIP(c, hipMemcpyAsync(user, c->d_user, n * 4, hipMemcpyDeviceToHost, c->stream));
        if (disc) PIE_HIP(c, hipMemcpyAsync(disc, c->d_disc, n * 4, hipMemcpyDeviceToHost, c->stream));
    }
    PIE_HIP(c, hipStreamSynchronize(c->stream));
    return PIE_OK;
}

int pie_set_end(pie_ctx* c, const int32_t* rows, const int64_t* new_end, size_t k)
{
    if (!c) return PIE_E_INVAL;
    if (k == 0) return PIE_OK;
    if (!rows || !new_end) return fail(c, PIE_E_INVAL, "NULL pointer");
    if (c->n_flight || c->b_flight) return fail(c, PIE_E_STATE, "table change while a scan is in flight");
    for (size_t i = 0; i < k; ++i)
        if (rows[i] < 0 || rows[i] >= c->n) return fail(c, PIE_E_INVAL, "row %d outside the table", rows[i]);
    PIE_HIP(c, hipSetDevice(c->device));
    if (c->async_mutations && !c->ord.valid) { // queued, not waited for (see AsyncStage); the rows were checked above
        pie_ctx::AsyncStage* a = nullptr;
        int rca = async_stage(c, k * 12 + 64, &a);
        if (rca) return rca;
        memcpy(a->h, new_end, k * 8);
        memcpy(a->h + k * 8, rows, k * 4);
        PIE_HIP(c, hipMemcpyAsync(a->d, a->h, k * 12, hipMemcpyHostToDevice, c->stream));
        hipLaunchKernelGGL(k_set_end, dim3((unsigned)((k + 255) / 256)), dim3(256), 0, c->stream, c->d_end,
                           reinterpret_cast<const int*>(a->d + k * 8), reinterpret_cast<const long long*>(a->d), (long long)k, c->n,
                           c->d_key, c->key_base, c->key_shift, c->d_fkey, c->fkey_base, c->fkey_shift, ord_mirror_of(c));
        PIE_HIP(c, hipGetLastError());
        PIE_HIP(c, hipEventRecord(a->ev, c->stream));
        a->pending = true;
        c->key_dirty = true;
        return PIE_OK;
    }
    // staged like the append path: one upload of [new_end k | rows k], one kernel (end + both keys), one wait
    int rc = ensure_stage(c, k * 12 + 64);
    if (rc) return rc;
    memcpy(c->h_stage, new_end, k * 8);
    memcpy(c->h_stage + k * 8, rows, k * 4);
    PIE_HIP(c, hipMemcpyAsync(c->d_stage, c->h_stage, k * 12, hipMemcpyHostToDevice, c->stream));
    hipLaunchKernelGGL(k_set_end, dim3((unsigned)((k + 255) / 256)), dim3(256), 0, c->stream, c->d_end,
                       reinterpret_cast<const int*>(c->d_stage + k * 8), reinterpret_cast<const long long*>(c->d_stage), (long long)k, c->n,
                       c->d_key, c->key_base, c->key_shift, c->d_fkey, c->fkey_base, c->fkey_shift, ord_mirror_of(c));
    PIE_HIP(c, hipGetLastError());
    c->key_dirty = true;
    PIE_HIP(c, hipStreamSynchronize(c->stream));
    if (c->ord.h_stale && *(volatile unsigned int*)c->ord.h_stale) { // a row the run does not hold is live again
        *c->ord.h_stale = 0;
        ord_invalidate(c);
    }
    return PIE_OK;
}

int pie_delete_user(pie_ctx* c, int32_t user, int32_t* rows_out, size_t cap, size_t* n_deleted)
{
    if (!c) return PIE_E_INVAL;
    if (n_deleted) *n_deleted = 0;
    if (c->n == 0 || user < 0 || user >= c->n_users) return PIE_OK; // unknown / falsy id: no-op (sessionStore.js:56-58)
    // the rows are tombstoned even when rows_out is too small to list them (PIE_E_CAPACITY then tells the count)
    return run_row_list<1>(c, (long long)user, 0, rows_out, cap, n_deleted);
}

int pie_prune_before(pie_ctx* c, int64_t cutoff, int32_t* rows_out, size_t cap, size_t* n_pruned)
{
    if (!c) return PIE_E_INVAL;
    if (n_pruned) *n_pruned = 0;
    if (c->n == 0) return PIE_OK;
    return run_row_list<2>(c, (long long)cutoff, 0, rows_out, cap, n_pruned);
}

int pie_retention_purge(pie_ctx* c, int64_t now, int32_t months, int64_t tz_offset_ms, int32_t* rows_out, size_t cap,
                        size_t* n_purged)
{
    if (!c) return PIE_E_INVAL;
    if (n_purged) *n_purged = 0;
    if (months < -32768 || months > 32767) return fail(c, PIE_E_INVAL, "months outside int16");
    if (tz_offset_ms % 60000 != 0 || tz_offset_ms > 86400000LL || tz_offset_ms < -86400000LL)
        return fail(c, PIE_E_INVAL, "tz offset must be whole minutes within a day");
    if (c->n == 0) return PIE_OK;
    const long long packed = ((tz_offset_ms / 60000) << 16) | (long long)((unsigned)months & 0xFFFFu);
    return run_row_list<3>(c, (long long)now, packed, rows_out, cap, n_purged);
}

int pie_retention_purge_tz(pie_ctx* c, int64_t now, int32_t months, const int64_t* transitions_utc_ms, const int64_t* offsets_ms,
                           int32_t n_transitions, int32_t* rows_out, size_t cap, size_t* n_purged)
{
    if (!c) return PIE_E_INVAL;
    if (n_purged) *n_purged = 0;
    if (months < -32768 || months > 32767) return fail(c, PIE_E_INVAL, "months outside int16");
    if (n_transitions < 0 || n_transitions > 65536 || !offsets_ms || (n_transitions > 0 && !transitions_utc_ms))
        return fail(c, PIE_E_INVAL, "bad transition table");
    const int n = n_transitions;
    for (int i = 0; i <= n; ++i)
        if (offsets_ms[i] > 2 * 86400000LL || offsets_ms[i] < -2 * 86400000LL) return fail(c, PIE_E_INVAL, "offset %d outside two days", i);
    // transitions ascending, and ascending on the old offset's local clock too (what the local -> UTC search relies on)
    for (int i = 1; i < n; ++i)
        if (transitions_utc_ms[i] <= transitions_utc_ms[i - 1] || transitions_utc_ms[i] + offsets_ms[i] <= transitions_utc_ms[i - 1] + offsets_ms[i - 1])
            return fail(c, PIE_E_INVAL, "transitions must ascend (entry %d)", i);
    if (c->n == 0) return PIE_OK;
    PIE_HIP(c, hipSetDevice(c->device));
    // the table on the device: [n, months, T[n], L[n], off[n + 1]] through the context's staging block
    const size_t words = 2 + (size_t)n * 2 + (size_t)n + 1;
    if (c->n_flight || c->b_flight) return fail(c, PIE_E_STATE, "a scan is in flight");
    int rc = sync_all(c); // the staging block is shared with appends / touches
    if (rc) return rc;
    rc = ensure_stage(c, words * 8);
    if (rc) return rc;
    long long* h = reinterpret_cast<long long*>(c->h_stage);
    h[0] = n;
    h[1] = months;
    for (int i = 0; i < n; ++i) {
        h[2 + i] = transitions_utc_ms[i];
        h[2 + n + i] = transitions_utc_ms[i] + offsets_ms[i];
    }
    for (int i = 0; i <= n; ++i) h[2 + 2 * n + i] = offsets_ms[i];
    PIE_HIP(c, hipMemcpyAsync(c->d_stage, c->h_stage, words * 8, hipMemcpyHostToDevice, c->stream));
    return run_row_list<4>(c, (long long)now, (long long)(uintptr_t)c->d_stage, rows_out, cap, n_purged);
}

int pie_set_disciplines(pie_ctx* c, uint64_t mask, int32_t n_disc)
{
    if (!c) return PIE_E_INVAL;
    if (n_disc < 0 || n_disc > 64) return fail(c, PIE_E_INVAL, "n_disc %d outside [0, 64]", n_disc);
    c->disc_mask = mask;
    c->n_disc = n_disc;
    return PIE_OK;
}

int pie_scan_device(pie_ctx* c, int64_t now, int64_t cutoff, size_t* m_out)
{
    if (!c) return PIE_E_INVAL;
    if (c->b_flight) return fail(c, PIE_E_STATE, "a batch is in flight: finish it first");
    PIE_HIP(c, hipSetDevice(c->device));
    int rc = run_scan(c, now, cutoff);
    if (m_out) *m_out = c->res ? (size_t)c->res->last.m : 0;
    return rc;
}

int pie_scan_begin(pie_ctx* c, int64_t now, int64_t cutoff)
{
    if (!c) return PIE_E_INVAL;
    if (c->b_flight) return fail(c, PIE_E_STATE, "a batch is in flight: finish it first");
    PIE_HIP(c, hipSetDevice(c->device));
    return scan_begin(c, now, cutoff);
}

int pie_scan_finish(pie_ctx* c, size_t* m_out)
{
    if (!c) return PIE_E_INVAL;
    PIE_HIP(c, hipSetDevice(c->device));
    int rc = scan_finish(c);
    if (m_out) *m_out = c->res ? (size_t)c->res->last.m : 0;
    return rc;
}

int pie_scan_begin_packed(pie_ctx* c, int64_t now, int64_t cutoff, void* dst_i32, size_t u_pad, size_t idx_cap)
{
    if (!c) return PIE_E_INVAL;
    if (c->b_flight) return fail(c, PIE_E_STATE, "a batch is in flight: finish it first");
    if (!dst_i32 || u_pad < (size_t)c->n_users || u_pad > 0x7FFFFFF0u) return fail(c, PIE_E_INVAL, "bad message destination / u_pad < n_users");
    PIE_HIP(c, hipSetDevice(c->device));
    return scan_begin(c, now, cutoff, (int*)dst_i32, (int)u_pad, (long long)idx_cap);
}

int pie_scan_begin_packed2(pie_ctx* c, int64_t now, int64_t cutoff, void* dst_i32, size_t u_pad, size_t idx_cap, void* counts_dst_i32)
{
    if (!c) return PIE_E_INVAL;
    if (c->b_flight) return fail(c, PIE_E_STATE, "a batch is in flight: finish it first");
    if (!dst_i32 || u_pad < (size_t)c->n_users || u_pad > 0x7FFFFFF0u) return fail(c, PIE_E_INVAL, "bad message destination / u_pad < n_users");
    PIE_HIP(c, hipSetDevice(c->device));
    return scan_begin(c, now, cutoff, (int*)dst_i32, (int)u_pad, (long long)idx_cap, (int*)counts_dst_i32);
}

int pie_scan_finish_packed(pie_ctx* c, size_t* m_out, int* ready_out)
{
    if (!c) return PIE_E_INVAL;
    if (ready_out) *ready_out = 0;
    PIE_HIP(c, hipSetDevice(c->device));
    Slot* slp = oldest_in_flight(c);
    if (!slp) return fail(c, PIE_E_STATE, "pie_scan_finish_packed without pie_scan_begin_packed");
    if (!slp->msg) return fail(c, PIE_E_STATE, "the oldest scan in flight was not begun with pie_scan_begin_packed");
    int rc = scan_finish(c);
    if (rc) return rc;
    Slot& sl = *c->res;
    if (m_out) *m_out = (size_t)sl.last.m;
    // K2 wrote the whole message iff it ran the fused form and no bucket outgrew the direct slots
    if (sl.msg_by_k2 && sl.last.n_small + sl.last.n_seg == 0) {
        if (ready_out) *ready_out = 1;
        return PIE_OK;
    }
    // the unfused K2 wrote neither destination; the fused one wrote the counts but not the rows of the outgrown buckets
    if (sl.msg_counts && !sl.msg_by_k2)
        PIE_HIP(c, hipMemcpyAsync(sl.msg_counts, sl.counts_ord, (size_t)c->n_users * 4, hipMemcpyDefault, c->stream));
    return pie_pack_results_device(c, sl.msg, (size_t)sl.msg_u_pad, (size_t)sl.msg_cap);
}

int pie_scan_batch_begin(pie_ctx* c, const pie_query* queries, int n_q)
{
    if (!c) return PIE_E_INVAL;
    PIE_HIP(c, hipSetDevice(c->device));
    return batch_begin(c, queries, n_q, 0, nullptr, 0, 0, 0, nullptr, 0);
}

int pie_scan_batch_begin_packed(pie_ctx* c, const pie_query* queries, int n_q, void* msg_i32, size_t msg_stride_words, size_t u_pad,
                                size_t idx_cap, void* counts_i32, size_t counts_stride_words)
{
    if (!c) return PIE_E_INVAL;
    if (!msg_i32 || u_pad < (size_t)c->n_users || u_pad > 0x7FFFFFF0u || msg_stride_words < u_pad + 2 + idx_cap)
        return fail(c, PIE_E_INVAL, "bad message destination / u_pad < n_users / stride below u_pad + 2 + idx_cap");
    if (counts_i32 && counts_stride_words < (size_t)c->n_users) return fail(c, PIE_E_INVAL, "counts stride below n_users");
    PIE_HIP(c, hipSetDevice(c->device));
    return batch_begin(c, queries, n_q, 1, (int*)msg_i32, (long long)msg_stride_words, (int)u_pad, (long long)idx_cap, (int*)counts_i32,
                       (long long)counts_stride_words);
}

int pie_scan_batch_begin_union(pie_ctx* c, const pie_query* queries, int n_q, void* msg_i32, size_t u_pad, size_t cap)
{
    if (!c) return PIE_E_INVAL;
    if (!msg_i32 || u_pad < (size_t)c->n_users || u_pad > 0x7FFFFFF0u) return fail(c, PIE_E_INVAL, "bad union destination / u_pad < n_users");
    PIE_HIP(c, hipSetDevice(c->device));
    return batch_begin(c, queries, n_q, 2, (int*)msg_i32, 0, (int)u_pad, (long long)cap, nullptr, 0);
}

// No further begin is coming for now: the tails that were waiting for a ride in their lane's next launch are queued at once,
// side by side, instead of one by one as their batches are finished (the end of a burst: 20 batches on three lanes drained
// through three tail launches in a row).
int pie_scan_batch_flush(pie_ctx* c)
{
    if (!c) return PIE_E_INVAL;
    PIE_HIP(c, hipSetDevice(c->device));
    // Oldest first, every batch of the lane whose tail still waits — not only the youngest: a batch that only falls back queues no
    // launch, so the batch before it keeps its tail until its finish, and tails must run in the order of their batches (a tail
    // zeroes the span of the batch after the next: found by the differential fuzz, which flushed the third batch of a lane
    // while the first one's tail was still waiting behind such a batch).
    for (int lane = 0; lane < kLaneMax; ++lane) {
        for (int age = c->lane_flight[lane]; age >= 1; --age) {
            BatchSlot& b = c->bslot[lane * kBatchSlots + (c->lane_next[lane] + kBatchSlots - age) % kBatchSlots];
            if (b.in_flight && b.k2_pending && !b.ordered && !b.unsupported) launch_batch_k2(c, b, b.stream ? b.stream : c->stream);
        }
    }
    PIE_HIP(c, hipGetLastError());
    return PIE_OK;
}

int pie_scan_batch_finish_packed(pie_ctx* c, size_t* m_out, int* ready_out)
{
    if (!c) return PIE_E_INVAL;
    PIE_HIP(c, hipSetDevice(c->device));
    int rc = batch_finish(c, ready_out);
    if (rc) return rc;
    if (m_out)
        for (int q = 0; q < c->bres->n_q; ++q) m_out[q] = (size_t)c->bres->last[q].m;
    return PIE_OK;
}

int pie_scan_batch_finish(pie_ctx* c, size_t* m_out) { return pie_scan_batch_finish_packed(c, m_out, nullptr); }

int pie_scan_batch(pie_ctx* c, const pie_query* queries, int n_q, size_t* m_out)
{
    if (!c) return PIE_E_INVAL;
    if (c->b_flight) return fail(c, PIE_E_STATE, "a batch is in flight: finish it first");
    int rc = pie_scan_batch_begin(c, queries, n_q);
    if (rc) return rc;
    return pie_scan_batch_finish(c, m_out);
}

int pie_batch_union_device_ptrs(pie_ctx* c, void** uoff_dev, void** rows_dev, void** mask_lo_dev, void** mask_hi_dev, size_t* mu_out)
{
    if (!c) return PIE_E_INVAL;
    if (uoff_dev) *uoff_dev = nullptr;
    if (rows_dev) *rows_dev = nullptr;
    if (mask_lo_dev) *mask_lo_dev = nullptr;
    if (mask_hi_dev) *mask_hi_dev = nullptr;
    if (mu_out) *mu_out = 0;
    if (!c->bres || !c->bres->have_result) return fail(c, PIE_E_STATE, "no batch result on this context");
    if (int rc_o = order_after_batch(c, *c->bres)) return rc_o; // the batch may have run on another lane's stream
    const BatchSlot& b = *c->bres;
    if (!b.union_ok) return PIE_OK; // queries fell back / the batch ran on the ordered run: per-query lists only
    if (uoff_dev) *uoff_dev = b.uoff;
    if (rows_dev) *rows_dev = b.urows;
    if (mask_lo_dev) *mask_lo_dev = b.umlo;
    if (mask_hi_dev) *mask_hi_dev = b.n_q > 32 ? b.umhi : nullptr;
    if (mu_out) *mu_out = (size_t)b.mu;
    return PIE_OK;
}

int pie_batch_read_union(pie_ctx* c, int64_t* uoff_out, int32_t* rows_out, uint64_t* masks_out, size_t cap, size_t* mu_out)
{
    if (!c) return PIE_E_INVAL;
    if (mu_out) *mu_out = 0;
    if (!c->bres || !c->bres->have_result) return fail(c, PIE_E_STATE, "no batch result on this context");
    if (int rc_o = order_after_batch(c, *c->bres)) return rc_o; // the batch may have run on another lane's stream
    const BatchSlot& b = *c->bres;
    if (!b.union_ok) return fail(c, PIE_E_STATE, "this batch has no union result (queries fell back to the general path, or it ran on the ordered run): read the per-query results");
    PIE_HIP(c, hipSetDevice(c->device));
    hipStream_t a = c->stream;
    const size_t mu = (size_t)b.mu;
    if (mu_out) *mu_out = mu;
    if (uoff_out) PIE_HIP(c, hipMemcpyAsync(uoff_out, b.uoff, ((size_t)c->n_users + 1) * 8, hipMemcpyDeviceToHost, a));
    if ((rows_out || masks_out) && mu > cap) {
        PIE_HIP(c, hipStreamSynchronize(a));
        return fail(c, PIE_E_CAPACITY, "cap %zu < union rows %zu", cap, mu);
    }
    if (rows_out && mu) PIE_HIP(c, hipMemcpyAsync(rows_out, b.urows, mu * 4, hipMemcpyDeviceToHost, a));
    std::vector<uint32_t> lo, hi;
    if (masks_out && mu) {
        lo.resize(mu);
        PIE_HIP(c, hipMemcpyAsync(lo.data(), b.umlo, mu * 4, hipMemcpyDeviceToHost, a));
        if (b.n_q > 32) {
            hi.resize(mu);
            PIE_HIP(c, hipMemcpyAsync(hi.data(), b.umhi, mu * 4, hipMemcpyDeviceToHost, a));
        }
    }
    PIE_HIP(c, hipStreamSynchronize(a));
    if (masks_out)
        for (size_t i = 0; i < mu; ++i) masks_out[i] = (uint64_t)lo[i] | (hi.empty() ? 0ull : ((uint64_t)hi[i] << 32));
    return PIE_OK;
}

int pie_batch_result_device_ptrs(pie_ctx* c, int qi, void** counts_dev, void** offsets_dev, void** idx_dev)
{
    if (!c) return PIE_E_INVAL;
    if (!c->bres || !c->bres->have_result) return fail(c, PIE_E_STATE, "no batch result on this context");
    if (int rc_o = order_after_batch(c, *c->bres)) return rc_o; // the batch may have run on another lane's stream
    if (qi < 0 || qi >= c->bres->n_q) return fail(c, PIE_E_INVAL, "query %d outside the batch of %d", qi, c->bres->n_q);
    PIE_HIP(c, hipSetDevice(c->device));
    int rc = batch_need_list(c, *c->bres, qi);
    if (rc) return rc;
    const long long us = batch_users_stride(c);
    if (counts_dev) *counts_dev = c->bres->counts_ord + (long long)qi * us;
    if (offsets_dev) *offsets_dev = c->bres->offsets + (long long)qi * us;
    if (idx_dev) *idx_dev = c->bres->idx_of[qi];
    return PIE_OK;
}

int pie_batch_read_user_feed(pie_ctx* c, int qi, int32_t user, int32_t* idx_out, size_t idx_cap, size_t* k_out)
{
    if (!c) return PIE_E_INVAL;
    if (k_out) *k_out = 0;
    if (!c->bres || !c->bres->have_result) return fail(c, PIE_E_STATE, "no batch result on this context");
    if (int rc_o = order_after_batch(c, *c->bres)) return rc_o; // the batch may have run on another lane's stream
    BatchSlot& b = *c->bres;
    if (qi < 0 || qi >= b.n_q) return fail(c, PIE_E_INVAL, "query %d outside the batch of %d", qi, b.n_q);
    if (user < 0 || user >= c->n_users) return PIE_OK;
    PIE_HIP(c, hipSetDevice(c->device));
    if (b.union_ok && !b.list_ok[qi]) {
        // straight from the union: the user's (few) union rows, filtered by the query's bit — no per-query list is ever built
        long long off[2] = {0, 0};
        PIE_HIP(c, hipMemcpyAsync(off, b.uoff + user, sizeof off, hipMemcpyDeviceToHost, c->stream));
        PIE_HIP(c, hipStreamSynchronize(c->stream));
        const size_t ku = (size_t)(off[1] - off[0]);
        if (ku == 0) return PIE_OK;
        int rows_s[1 << kUnionShiftMax];
        unsigned masks_s[1 << kUnionShiftMax];
        std::vector<int> rows_h;
        std::vector<unsigned> masks_h;
        int* rows = rows_s;
        unsigned* masks = masks_s;
        if (ku > (size_t)(1 << kUnionShiftMax)) { // a batch on the ordered run: a head user's union is as long as its live rows
            rows_h.resize(ku);
            masks_h.resize(ku);
            rows = rows_h.data();
            masks = masks_h.data();
        }
        PIE_HIP(c, hipMemcpyAsync(rows, b.urows + off[0], ku * 4, hipMemcpyDeviceToHost, c->stream));
        PIE_HIP(c, hipMemcpyAsync(masks, (qi >= 32 ? b.umhi : b.umlo) + off[0], ku * 4, hipMemcpyDeviceToHost, c->stream));
        PIE_HIP(c, hipStreamSynchronize(c->stream));
        size_t k = 0;
        for (size_t i = 0; i < ku; ++i) k += (masks[i] >> (qi & 31)) & 1u;
        if (k_out) *k_out = k;
        if (k == 0) return PIE_OK;
        if (!idx_out || k > idx_cap) return fail(c, PIE_E_CAPACITY, "idx_cap %zu < feed length %zu", idx_cap, k);
        size_t at = 0;
        for (size_t i = 0; i < ku; ++i)
            if ((masks[i] >> (qi & 31)) & 1u) idx_out[at++] = rows[i];
        return PIE_OK;
    }
    void *dc = nullptr, *dof = nullptr, *di = nullptr;
    int rc = pie_batch_result_device_ptrs(c, qi, &dc, &dof, &di);
    if (rc) return rc;
    long long off[2] = {0, 0};
    PIE_HIP(c, hipMemcpyAsync(off, static_cast<long long*>(dof) + user, sizeof off, hipMemcpyDeviceToHost, c->stream));
    PIE_HIP(c, hipStreamSynchronize(c->stream));
    const size_t k = (size_t)(off[1] - off[0]);
    if (k_out) *k_out = k;
    if (k == 0) return PIE_OK;
    if (!idx_out || k > idx_cap) return fail(c, PIE_E_CAPACITY, "idx_cap %zu < feed length %zu", idx_cap, k);
    PIE_HIP(c, hipMemcpyAsync(idx_out, static_cast<int*>(di) + off[0], k * 4, hipMemcpyDeviceToHost, c->stream));
    PIE_HIP(c, hipStreamSynchronize(c->stream));
    return PIE_OK;
}

int pie_batch_fetch_requests(pie_ctx* c, const int32_t* qi, const int32_t* user, size_t n_req, size_t cap_rows, int64_t* req_off_out,
                             int32_t* idx_out, int64_t* start_out, int64_t* end_out, int32_t* disc_out, size_t* total_out)
{
    if (!c) return PIE_E_INVAL;
    if (total_out) *total_out = 0;
    if (!c->bres || !c->bres->have_result) return fail(c, PIE_E_STATE, "no batch result on this context");
    if (int rc_o = order_after_batch(c, *c->bres)) return rc_o; // the batch may have run on another lane's stream
    if (n_req == 0) return PIE_OK;
    if (!qi || !user || !req_off_out || n_req > (size_t)1 << 24) return fail(c, PIE_E_INVAL, "bad request list");
    BatchSlot& b = *c->bres;
    for (size_t i = 0; i < n_req; ++i)
        if (qi[i] < 0 || qi[i] >= b.n_q) return fail(c, PIE_E_INVAL, "request %zu: query %d outside the batch of %d", i, qi[i], b.n_q);
    PIE_HIP(c, hipSetDevice(c->device));
    hipStream_t s = c->stream;
    if (!b.union_ok || !c->d_pay) {
        // a batch without a union (queries fell back, the ordered run) or a table without the payload column: request by request
        size_t total = 0;
        std::vector<int32_t> tmp;
        for (size_t i = 0; i < n_req; ++i) {
            req_off_out[i] = (int64_t)total;
            size_t k = 0;
            int rc = pie_batch_read_user_feed(c, qi[i], user[i], nullptr, 0, &k);
            if (rc != PIE_OK && rc != PIE_E_CAPACITY) return rc;
            if (k && idx_out && total + k <= cap_rows) {
                rc = pie_batch_read_user_feed(c, qi[i], user[i], idx_out + total, k, &k);
                if (rc) return rc;
                rc = pie_fetch_rows(c, idx_out + total, k, start_out ? start_out + total : nullptr, end_out ? end_out + total : nullptr, nullptr,
                                    disc_out ? disc_out + total : nullptr);
                if (rc) return rc;
            }
            total += k;
        }
        req_off_out[n_req] = (int64_t)total;
        if (total_out) *total_out = total;
        if (idx_out && total > cap_rows) return fail(c, PIE_E_CAPACITY, "cap_rows %zu < %zu rows", cap_rows, total);
        return PIE_OK;
    }
    // device staging: [qi n | user n | cnt n | off (n + 1) x 8 | idx cap | disc cap | start cap x 8 | end cap x 8]
    const size_t n4 = ((n_req + 3) / 4) * 4, cap4 = ((cap_rows + 3) / 4) * 4;
    const size_t bytes = n4 * 12 + (n4 + 4) * 8 + cap4 * 24 + 64;
    int rc = ensure_stage(c, bytes);
    if (rc) return rc;
    char* h = c->h_stage;
    char* d = c->d_stage;
    memcpy(h, qi, n_req * 4);
    memcpy(h + n4 * 4, user, n_req * 4);
    PIE_HIP(c, hipMemcpyAsync(d, h, n4 * 8, hipMemcpyHostToDevice, s));
    int* d_qi = reinterpret_cast<int*>(d);
    int* d_us = reinterpret_cast<int*>(d + n4 * 4);
    int* d_cnt = reinterpret_cast<int*>(d + n4 * 8);
    long long* d_off = reinterpret_cast<long long*>(d + n4 * 12);
    int* d_idx = reinterpret_cast<int*>(d + n4 * 12 + (n4 + 4) * 8);
    int* d_disc = d_idx + cap4;
    long long* d_start = reinterpret_cast<long long*>(d_disc + cap4);
    long long* d_end = d_start + cap4;
    const unsigned blocks = (unsigned)((n_req + 255) / 256);
    const unsigned* hi = b.n_q > 32 ? b.umhi : b.umlo;
    hipLaunchKernelGGL(k_req_count, dim3(blocks), dim3(256), 0, s, (int)n_req, d_qi, d_us, c->n_users, b.uoff, b.umlo, hi, d_cnt);
    hipLaunchKernelGGL(k_block_prefix, dim3(1), dim3(256), 0, s, d_cnt, (int)n_req, d_off, (unsigned long long*)nullptr);
    hipLaunchKernelGGL(k_req_write, dim3(blocks), dim3(256), 0, s, (int)n_req, d_qi, d_us, c->n_users, b.uoff, b.urows, b.umlo, hi, d_off, c->d_pay,
                       c->d_end, (long long)cap_rows, d_idx, d_start, d_end, d_disc);
    PIE_HIP(c, hipGetLastError());
    PIE_HIP(c, hipMemcpyAsync(req_off_out, d_off, (n_req + 1) * 8, hipMemcpyDeviceToHost, s));
    PIE_HIP(c, hipStreamSynchronize(s));
    const size_t total = (size_t)req_off_out[n_req];
    if (total_out) *total_out = total;
    if (idx_out && total > cap_rows) return fail(c, PIE_E_CAPACITY, "cap_rows %zu < %zu rows", cap_rows, total);
    if (total && idx_out) {
        PIE_HIP(c, hipMemcpyAsync(idx_out, d_idx, total * 4, hipMemcpyDeviceToHost, s));
        if (disc_out) PIE_HIP(c, hipMemcpyAsync(disc_out, d_disc, total * 4, hipMemcpyDeviceToHost, s));
        if (start_out) PIE_HIP(c, hipMemcpyAsync(start_out, d_start, total * 8, hipMemcpyDeviceToHost, s));
        if (end_out) PIE_HIP(c, hipMemcpyAsync(end_out, d_end, total * 8, hipMemcpyDeviceToHost, s));
        PIE_HIP(c, hipStreamSynchronize(s));
    }
    return PIE_OK;
}

int pie_batch_read_results(pie_ctx* c, int qi, int32_t* counts_out, int64_t* offsets_out, int32_t* idx_out, size_t idx_cap, size_t* m_out)
{
    if (!c) return PIE_E_INVAL;
    void *dc = nullptr, *dof = nullptr, *di = nullptr;
    int rc = pie_batch_result_device_ptrs(c, qi, &dc, &dof, &di);
    if (rc) return rc;
    PIE_HIP(c, hipSetDevice(c->device));
    hipStream_t a = c->stream;
    const size_t m = (size_t)c->bres->last[qi].m;
    if (m_out) *m_out = m;
    if (counts_out) PIE_HIP(c, hipMemcpyAsync(counts_out, dc, (size_t)c->n_users * 4, hipMemcpyDeviceToHost, a));
    if (offsets_out) PIE_HIP(c, hipMemcpyAsync(offsets_out, dof, ((size_t)c->n_users + 1) * 8, hipMemcpyDeviceToHost, a));
    if (idx_out && m > idx_cap) {
        PIE_HIP(c, hipStreamSynchronize(a));
        return fail(c, PIE_E_CAPACITY, "idx_cap %zu < selected rows %zu", idx_cap, m);
    }
    if (idx_out && m) PIE_HIP(c, hipMemcpyAsync(idx_out, di, m * 4, hipMemcpyDeviceToHost, a));
    PIE_HIP(c, hipStreamSynchronize(a));
    return PIE_OK;
}

int pie_read_results(pie_ctx* c, int32_t* counts_out, int64_t* offsets_out, int32_t* idx_out, size_t idx_cap, size_t* m_out)
{
    if (!c) return PIE_E_INVAL;
    if (!c->res || !c->res->have_result) return fail(c, PIE_E_STATE, "no scan result on this context");
    PIE_HIP(c, hipSetDevice(c->device));
    Slot& sl = *c->res;
    hipStream_t a = c->stream;
    const size_t m = (size_t)sl.last.m;
    if (m_out) *m_out = m;
    if (counts_out) PIE_HIP(c, hipMemcpyAsync(counts_out, sl.counts_ord, (size_t)c->n_users * 4, hipMemcpyDeviceToHost, a));
    if (offsets_out) PIE_HIP(c, hipMemcpyAsync(offsets_out, sl.offsets, ((size_t)c->n_users + 1) * 8, hipMemcpyDeviceToHost, a));
    if (idx_out && m > idx_cap) {
        PIE_HIP(c, hipStreamSynchronize(a));
        return fail(c, PIE_E_CAPACITY, "idx_cap %zu < selected rows %zu", idx_cap, m);
    }
    if (idx_out && m) PIE_HIP(c, hipMemcpyAsync(idx_out, sl.out_idx, m * 4, hipMemcpyDeviceToHost, a));
    PIE_HIP(c, hipStreamSynchronize(a));
    return PIE_OK;
}

int pie_read_user_feed(pie_ctx* c, int32_t user, int32_t* idx_out, size_t idx_cap, size_t* k_out)
{
    if (!c) return PIE_E_INVAL;
    if (k_out) *k_out = 0;
    if (!c->res || !c->res->have_result) return fail(c, PIE_E_STATE, "no scan result on this context");
    if (user < 0 || user >= c->n_users) return PIE_OK;
    PIE_HIP(c, hipSetDevice(c->device));
    Slot& sl = *c->res;
    long long off[2] = {0, 0};
    PIE_HIP(c, hipMemcpyAsync(off, sl.offsets + user, sizeof off, hipMemcpyDeviceToHost, c->stream));
    PIE_HIP(c, hipStreamSynchronize(c->stream));
    const size_t k = (size_t)(off[1] - off[0]);
    if (k_out) *k_out = k;
    if (k == 0) return PIE_OK;
    if (!idx_out || k > idx_cap) return fail(c, PIE_E_CAPACITY, "idx_cap %zu < feed length %zu", idx_cap, k);
    PIE_HIP(c, hipMemcpyAsync(idx_out, sl.out_idx + off[0], k * 4, hipMemcpyDeviceToHost, c->stream));
    PIE_HIP(c, hipStreamSynchronize(c->stream));
    return PIE_OK;
}

int pie_scan(pie_ctx* c, int64_t now, int64_t cutoff, int32_t* counts_out, int64_t* offsets_out, int32_t* idx_out,
             size_t idx_cap, size_t* m_out)
{
    if (!c) return PIE_E_INVAL;
    if (c->b_flight) return fail(c, PIE_E_STATE, "a batch is in flight: finish it first");
    PIE_HIP(c, hipSetDevice(c->device));
    int rc = run_scan(c, now, cutoff);
    if (m_out) *m_out = c->res ? (size_t)c->res->last.m : 0;
    if (rc) return rc;
    return pie_read_results(c, counts_out, offsets_out, idx_out, idx_cap, m_out);
}

int pie_result_device_ptrs(pie_ctx* c, void** counts_dev, void** offsets_dev, void** idx_dev)
{
    if (!c) return PIE_E_INVAL;
    if (!c->res || !c->res->have_result) return fail(c, PIE_E_STATE, "no scan result on this context");
    if (counts_dev) *counts_dev = c->res->counts_ord;
    if (offsets_dev) *offsets_dev = c->res->offsets;
    if (idx_dev) *idx_dev = c->res->out_idx;
    return PIE_OK;
}

int pie_copy_results_device(pie_ctx* c, void* counts_dst, void* offsets_dst, void* idx_dst, size_t idx_cap)
{
    if (!c) return PIE_E_INVAL;
    if (!c->res || !c->res->have_result) return fail(c, PIE_E_STATE, "no scan result on this context");
    PIE_HIP(c, hipSetDevice(c->device));
    Slot& sl = *c->res;
    hipStream_t a = c->stream;
    if (counts_dst) PIE_HIP(c, hipMemcpyAsync(counts_dst, sl.counts_ord, (size_t)c->n_users * 4, hipMemcpyDeviceToDevice, a));
    if (offsets_dst)
        PIE_HIP(c, hipMemcpyAsync(offsets_dst, sl.offsets, ((size_t)c->n_users + 1) * 8, hipMemcpyDeviceToDevice, a));
    size_t m = (size_t)sl.last.m;
    if (m > idx_cap) m = idx_cap;
    if (idx_dst && m) PIE_HIP(c, hipMemcpyAsync(idx_dst, sl.out_idx, m * 4, hipMemcpyDeviceToDevice, a));
    return PIE_OK;
}

int pie_pack_results_device(pie_ctx* c, void* dst_i32, size_t u_pad, size_t idx_cap)
{
    if (!c) return PIE_E_INVAL;
    if (!c->res || !c->res->have_result) return fail(c, PIE_E_STATE, "no scan result on this context");
    if (!dst_i32 || u_pad < (size_t)c->n_users) return fail(c, PIE_E_INVAL, "bad pack destination / u_pad < n_users");
    PIE_HIP(c, hipSetDevice(c->device));
    Slot& sl = *c->res;
    const size_t total = u_pad + 2 + ((size_t)sl.last.m < idx_cap ? (size_t)sl.last.m : idx_cap);
    size_t grid = (total + 255) / 256;
    if (grid > (size_t)c->n_cus * 8) grid = (size_t)c->n_cus * 8;
    hipLaunchKernelGGL(k_pack_results, dim3((unsigned)grid), dim3(256), 0, c->stream, sl.offsets, c->n_users, (int)u_pad,
                       sl.sum, sl.out_idx, (long long)idx_cap, (int*)dst_i32);
    PIE_HIP(c, hipGetLastError());
    return PIE_OK;
}

int pie_batch_pack_union_device(pie_ctx* c, void* dst_i32, size_t u_pad, size_t cap)
{
    if (!c) return PIE_E_INVAL;
    if (!c->bres || !c->bres->have_result) return fail(c, PIE_E_STATE, "no batch result on this context");
    if (int rc_o = order_after_batch(c, *c->bres)) return rc_o; // the batch may have run on another lane's stream
    if (!dst_i32 || u_pad < (size_t)c->n_users || u_pad > 0x7FFFFFF0u) return fail(c, PIE_E_INVAL, "bad union destination / u_pad < n_users");
    PIE_HIP(c, hipSetDevice(c->device));
    return batch_pack_union(c, *c->bres, dst_i32, u_pad, cap);
}

int pie_fetch_rows(pie_ctx* c, const int32_t* idx, size_t m, int64_t* start, int64_t* end, int32_t* user, int32_t* disc)
{
    if (!c) return PIE_E_INVAL;
    if (m == 0) return PIE_OK;
    if (!idx) return fail(c, PIE_E_INVAL, "idx is NULL");
    if (c->n == 0) return fail(c, PIE_E_STATE, "no table loaded");
    PIE_HIP(c, hipSetDevice(c->device));
    // device scratch: [start m*8][end m*8][user m*4][disc m*4][idx m*4]
    char* d = nullptr;
    const size_t bytes = m * 28 + 64;
    PIE_HIP(c, hipMalloc(&d, bytes));
    long long* o_s = reinterpret_cast<long long*>(d);
    long long* o_e = o_s + m;
    int* o_u = reinterpret_cast<int*>(o_e + m);
    int* o_d = o_u + m;
    int* d_idx = o_d + m;
    hipError_t e = hipMemcpyAsync(d_idx, idx, m * 4, hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess) {
        hipLaunchKernelGGL(k_fetch_rows, dim3((unsigned)((m + 255) / 256)), dim3(256), 0, c->stream, d_idx, (long long)m,
                           c->n, c->d_start, c->d_end, c->d_user, c->d_disc, o_s, o_e, o_u, o_d);
        e = hipGetLastError();
    }
    if (e == hipSuccess && start) e = hipMemcpyAsync(start, o_s, m * 8, hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess && end) e = hipMemcpyAsync(end, o_e, m * 8, hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess && user) e = hipMemcpyAsync(user, o_u, m * 4, hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess && disc) e = hipMemcpyAsync(disc, o_d, m * 4, hipMemcpyDeviceToHost, c->stream);
    hipError_t e2 = hipStreamSynchronize(c->stream);
    (void)hipFree(d);
    if (e != hipSuccess || e2 != hipSuccess)
        return fail(c, PIE_E_HIP, "pie_fetch_rows: %s", hipGetErrorString(e != hipSuccess ? e : e2));
    return PIE_OK;
}

int pie_expired_queue(pie_ctx* c, int64_t prev_now, int64_t now, int32_t* queue_out, size_t cap, size_t* q_out)
{
    if (!c) return PIE_E_INVAL;
    if (q_out) *q_out = 0;
    if (c->cap_rows == 0) return fail(c, PIE_E_STATE, "no table loaded");
    if (c->n == 0) return PIE_OK;
    if (c->n_flight || c->b_flight) return fail(c, PIE_E_STATE, "a scan is in flight");
    if (getenv("PIE_EXPIRED_TWO_PASS")) // the count / prefix / write form, kept for A-B runs
        return run_row_list<0>(c, (long long)prev_now, (long long)now, queue_out, cap, q_out);
    PIE_HIP(c, hipSetDevice(c->device));
    hipStream_t s = c->stream;
    // workspace: slot 0's out_idx is the device-side queue, slot 1's out_idx the per-wave staging, slot 0's
    // blk_count the per-wave counts (4 per block of the liveness-first plan: <= the streaming plan's block count)
    Slot& q = c->slot[0];
    Slot& st = c->slot[1];
    q.have_result = st.have_result = false;
    c->res = nullptr;
    const int blocks = c->plan_blocks[1];
    const long long rpb = c->plan_rows[1];
    const int n_waves = blocks * kK1Waves;
    const bool prof = c->profiling && c->ring_used < kEventRing;
    if (prof && (int)c->ring.size() <= c->ring_used) {
        ScanEvents e{};
        PIE_HIP(c, hipEventCreate(&e.e0));
        PIE_HIP(c, hipEventCreate(&e.e1));
        PIE_HIP(c, hipEventCreate(&e.e2));
        c->ring.push_back(e);
    }
    if (prof) PIE_HIP(c, hipEventRecord(c->ring[c->ring_used].e0, s));
    if (c->key_ok && c->keyed_enabled && !getenv("PIE_EXPIRED_ON_END"))
        hipLaunchKernelGGL(k_expired_stage_keyed<2>, dim3(blocks), dim3(kK1Threads), 0, s, c->d_key, c->d_end, c->n, rpb,
                           (long long)prev_now, (long long)now, host_key_of(c, prev_now), host_key_of(c, now), st.out_idx, q.blk_count);
    else
        hipLaunchKernelGGL(k_expired_stage<8>, dim3(blocks), dim3(kK1Threads), 0, s, c->d_end, c->n, rpb, (long long)prev_now,
                           (long long)now, st.out_idx, q.blk_count);
    if (prof) PIE_HIP(c, hipEventRecord(c->ring[c->ring_used].e1, s));
    // the prefix kernel hands the queue's length to the host itself (mapped memory, seq last): no copy, no stream drain
    const unsigned long long seq = ++c->seq_counter;
    hipLaunchKernelGGL(k_block_prefix_wide, dim3(1), dim3(1024), 0, s, q.blk_count, n_waves, c->d_blk_off, &c->d_summary->m,
                       c->expired_copy_total ? (HostSummary*)nullptr : q.h_sum_dev, seq);
    hipLaunchKernelGGL(k_expired_gather, dim3(blocks < c->n_cus * 8 ? blocks : c->n_cus * 8), dim3(256), 0, s, st.out_idx,
                       q.blk_count, c->d_blk_off, n_waves, rpb / kK1Waves, q.out_idx, c->cap_rows);
    PIE_HIP(c, hipGetLastError());
    if (prof) {
        PIE_HIP(c, hipEventRecord(c->ring[c->ring_used].e2, s));
        c->ring_used++;
    }
    size_t k = 0;
    if (c->expired_copy_total) {
        PIE_HIP(c, hipMemcpyAsync(c->h_summary, c->d_summary, sizeof(Summary), hipMemcpyDeviceToHost, s));
        PIE_HIP(c, hipStreamSynchronize(s));
        k = (size_t)c->h_summary->m;
    } else {
        // bounded wait like a scan's; the gather behind the prefix kernel may still run when the length is known (everything that
        // reads the queue afterwards is queued on this stream, behind it)
        volatile unsigned long long* hs = &q.h_sum->seq;
        unsigned long long spins = 0;
        timespec t0{};
        clock_gettime(CLOCK_MONOTONIC, &t0);
        while (*hs != seq) {
            __builtin_ia32_pause();
            if ((++spins & 0x3FFFF) == 0) {
                const hipError_t e = hipStreamQuery(s);
                timespec t1{};
                clock_gettime(CLOCK_MONOTONIC, &t1);
                const double waited_ms = (double)(t1.tv_sec - t0.tv_sec) * 1e3 + (double)(t1.tv_nsec - t0.tv_nsec) * 1e-6;
                if (e != hipSuccess && e != hipErrorNotReady) return fail(c, PIE_E_HIP, "expired queue failed: %s", hipGetErrorString(e));
                if (waited_ms > c->wait_deadline_ms || (e == hipSuccess && *hs != seq && waited_ms > 1000.0))
                    return fail(c, PIE_E_HIP, "expired-queue length not published within %.0f ms (PIE_WAIT_DEADLINE_MS): kernel hung?", waited_ms);
            }
        }
        __atomic_thread_fence(__ATOMIC_ACQUIRE);
        k = (size_t)q.h_sum->s.m;
    }
    if (q_out) *q_out = k;
    if (queue_out && k > cap) return fail(c, PIE_E_CAPACITY, "queue cap %zu < %zu", cap, k);
    if (queue_out && k) {
        PIE_HIP(c, hipMemcpyAsync(queue_out, q.out_idx, k * 4, hipMemcpyDeviceToHost, s));
        PIE_HIP(c, hipStreamSynchronize(s));
    }
    return PIE_OK;
}

// The reference's archive chain on the device: group stats -> threshold on the host (U values) -> group-qualified
// scan (buckets per group, rows in table order) -> groups laid out in first-appearance order.
// The reference's archive chain entirely on the device (round 3; the threshold and the group order used to go through the
// host, the selection through the general scan path with its histogram atomics and per-bucket order):
//   1. k_arch_flag        one pass over (start, end): which groups hold a live row with start <= now - window            16 B/row
//                         (<=> now - min(start) >= window, sqlProvider.js:798) — a bitmap; `user` only for rows that pass
//   2. k_arch_select x 2  the selection, order-preserving (count per block, prefix, write): (group, row) pairs in table
//                         order, the bitmap looked up in LDS                                                          2 x 12 B/row
//   3. stable radix sort of the M pairs by group: the rows of a group stay in table order, and the head of its run IS its
//      first row — the first-appearance order (Map insertion order, :769-789) costs no per-row work
//   4. per group: key = first row; radix sort of the U (key, group) pairs; sizes in that order; exclusive scan; gather of
//      every group's run into its place in the queue
// The sorts and the scan are rocPRIM's (plain library primitives, as hipBLASLt would be for a plain GEMM); everything that
// touches the table is above.  Algorithmic bytes: 20 B/row (start, end, user: the group statistics) + 12 B/row (end, user:
// the selection) + 4 B per queued row.
int pie_archive_queue(pie_ctx* c, int64_t now, int64_t window_ms, int32_t* queue_out, size_t cap, size_t* q_out)
{
    if (!c) return PIE_E_INVAL;
    if (q_out) *q_out = 0;
    if (c->cap_rows == 0) return fail(c, PIE_E_STATE, "no table loaded");
    if (c->n == 0) return PIE_OK;
    if (c->n_flight || c->b_flight) return fail(c, PIE_E_STATE, "a scan is in flight");
    PIE_HIP(c, hipSetDevice(c->device));
    int rc = sync_all(c);
    if (rc) return rc;
    rc = ensure_sel(c);
    if (rc) return rc;
    hipStream_t s = c->stream;
    const size_t U = (size_t)c->n_users;
    // per-group scratch: carved from one allocation kept with the context
    const size_t u4 = ((U + 63) / 64) * 64;
    const int words = (int)((U + 31) / 32);
    const size_t grp_bytes = u4 * 4 * 9 + 256;
    if (c->arch_bytes < grp_bytes) {
        dfree(c->d_arch);
        c->arch_bytes = 0;
        PIE_HIP(c, hipMalloc(&c->d_arch, grp_bytes));
        c->arch_bytes = grp_bytes;
    }
    char* p = c->d_arch;
    auto carve = [&](size_t bytes) { char* q = p; p += bytes; return q; };
    unsigned int* d_bits = reinterpret_cast<unsigned int*>(carve(u4 * 4)); // (words <= u4)
    int* d_ghead = reinterpret_cast<int*>(carve(u4 * 4));
    int* d_gfirst = reinterpret_cast<int*>(carve(u4 * 4));
    int* d_glast = reinterpret_cast<int*>(carve(u4 * 4));
    unsigned int* d_key[2] = {reinterpret_cast<unsigned int*>(carve(u4 * 4)), reinterpret_cast<unsigned int*>(carve(u4 * 4))};
    int* d_val[2] = {reinterpret_cast<int*>(carve(u4 * 4)), reinterpret_cast<int*>(carve(u4 * 4))};
    unsigned int* d_size = reinterpret_cast<unsigned int*>(carve(u4 * 4));
    unsigned int* d_nqual = reinterpret_cast<unsigned int*>(carve(256));
    Slot& sl = c->slot[0];
    Slot& other = c->slot[1];
    sl.have_result = other.have_result = false;
    c->res = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    if (c->profiling) {
        PIE_HIP(c, hipEventCreate(&ev0));
        PIE_HIP(c, hipEventCreate(&ev1));
        PIE_HIP(c, hipEventRecord(ev0, s));
    }
    // now - earliest >= window  <=>  earliest <= now - window, in 128 bits (JS numbers do not wrap)
    const __int128 lim128 = (__int128)now - (__int128)window_ms;
    const bool none = lim128 < (__int128)INT64_MIN;
    const long long limit = lim128 > (__int128)INT64_MAX ? INT64_MAX : (none ? INT64_MIN : (long long)lim128);
    PIE_HIP(c, hipMemsetAsync(d_bits, 0, (size_t)words * 4, s));
    PIE_HIP(c, hipMemsetAsync(d_nqual, 0, 4, s));
    hipLaunchKernelGGL(k_arch_flag, dim3(c->n_cus * 16), dim3(256), 0, s, c->d_start, c->d_end, c->d_user, c->n, c->n_users, limit, none, d_bits);
    hipLaunchKernelGGL(k_arch_popc, dim3((unsigned)((words + 255) / 256)), dim3(256), 0, s, d_bits, words, d_nqual);
    // the selection: (group, row) pairs in table order
    const int blocks = c->plan_blocks[0];
    const long long rpb = c->plan_rows[0]; // a multiple of the streaming tile: even
    const bool lds_bits = (size_t)words * 4 <= (size_t)48 * 1024; // up to 393 216 groups; beyond: the bitmap is read through the caches
    const size_t lds = lds_bits ? (size_t)words * 4 : 0;
    if (lds_bits) hipLaunchKernelGGL((k_arch_select<false, true>), dim3(blocks), dim3(256), lds, s, c->d_end, c->d_user, c->n, rpb, d_bits, words, c->n_users, sl.blk_count, (const long long*)nullptr, (unsigned int*)nullptr, (int*)nullptr);
    else hipLaunchKernelGGL((k_arch_select<false, false>), dim3(blocks), dim3(256), 0, s, c->d_end, c->d_user, c->n, rpb, d_bits, words, c->n_users, sl.blk_count, (const long long*)nullptr, (unsigned int*)nullptr, (int*)nullptr);
    hipLaunchKernelGGL(k_block_prefix, dim3(1), dim3(256), 0, s, sl.blk_count, blocks, c->d_blk_off, &c->d_summary->m);
    PIE_HIP(c, hipGetLastError());
    unsigned int n_qual = 0;
    PIE_HIP(c, hipMemcpyAsync(c->h_summary, c->d_summary, sizeof(Summary), hipMemcpyDeviceToHost, s));
    PIE_HIP(c, hipMemcpyAsync(&n_qual, d_nqual, 4, hipMemcpyDeviceToHost, s));
    PIE_HIP(c, hipStreamSynchronize(s));
    const size_t q = (size_t)c->h_summary->m;
    if (q_out) *q_out = q;
    c->arch_alg_bytes = 32ull * (unsigned long long)c->n + 4ull * q;
    if (queue_out && q > cap) return fail(c, PIE_E_CAPACITY, "queue cap %zu < %zu", cap, q);
    if (q) {
        // pair arrays in slot 0's record staging (16 B per row of capacity: four 4-byte arrays of q <= n entries); the
        // sorts' temporary storage and the queue in slot 1's
        unsigned int* k_in = reinterpret_cast<unsigned int*>(sl.sel);
        unsigned int* k_out = k_in + q;
        int* v_in = reinterpret_cast<int*>(k_out + q);
        int* v_out = v_in + q;
        int* d_queue = reinterpret_cast<int*>(other.sel);
        size_t need = 0;
        // temporary storage of the library primitives: a buffer of its own, grown to what they ask for (a table of a handful of
        // rows and 600 000 users wants more for the per-group sort than the table's record staging holds: found by the fuzz)
        auto tmp_for = [&](size_t bytes) -> void* {
            if (bytes > c->arch_tmp_bytes) {
                if (hipStreamSynchronize(s) != hipSuccess) return nullptr;
                dfree(c->d_arch_tmp);
                c->arch_tmp_bytes = 0;
                if (hipMalloc(&c->d_arch_tmp, bytes + bytes / 4 + 4096) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
                c->arch_tmp_bytes = bytes + bytes / 4 + 4096;
            }
            return c->d_arch_tmp;
        };
        void* tmp = nullptr;
        if (lds_bits) hipLaunchKernelGGL((k_arch_select<true, true>), dim3(blocks), dim3(256), lds, s, c->d_end, c->d_user, c->n, rpb, d_bits, words, c->n_users, (int*)nullptr, c->d_blk_off, k_in, v_in);
        else hipLaunchKernelGGL((k_arch_select<true, false>), dim3(blocks), dim3(256), 0, s, c->d_end, c->d_user, c->n, rpb, d_bits, words, c->n_users, (int*)nullptr, c->d_blk_off, k_in, v_in);
        unsigned g_bits = 1;
        while (g_bits < 32 && (1ull << g_bits) < (unsigned long long)U) ++g_bits;
        PIE_HIP(c, rocprim::radix_sort_pairs(nullptr, need, k_in, k_out, v_in, v_out, q, 0u, g_bits, s));
        if (!(tmp = tmp_for(need))) return fail(c, PIE_E_NOMEM, "sort scratch of %zu bytes", need);
        PIE_HIP(c, rocprim::radix_sort_pairs(tmp, need, k_in, k_out, v_in, v_out, q, 0u, g_bits, s));
        unsigned hb = (unsigned)((q + 255) / 256);
        if (hb > (unsigned)c->n_cus * 16) hb = (unsigned)c->n_cus * 16;
        hipLaunchKernelGGL(k_arch_heads, dim3(hb), dim3(256), 0, s, k_out, v_out, (long long)q, d_ghead, d_gfirst, d_glast);
        hipLaunchKernelGGL(k_arch_group_keys, dim3((unsigned)((U + 255) / 256)), dim3(256), 0, s, d_bits, d_gfirst, c->n_users, d_key[0], d_val[0]);
        PIE_HIP(c, rocprim::radix_sort_pairs(nullptr, need, d_key[0], d_key[1], d_val[0], d_val[1], U, 0u, 32u, s));
        if (!(tmp = tmp_for(need))) return fail(c, PIE_E_NOMEM, "sort scratch of %zu bytes", need);
        PIE_HIP(c, rocprim::radix_sort_pairs(tmp, need, d_key[0], d_key[1], d_val[0], d_val[1], U, 0u, 32u, s));
        hipLaunchKernelGGL(k_arch_sizes, dim3((unsigned)((n_qual + 255) / 256)), dim3(256), 0, s, d_val[1], d_nqual, d_ghead, d_glast, d_size);
        unsigned int* d_off = d_key[0]; // free again
        PIE_HIP(c, rocprim::exclusive_scan(nullptr, need, d_size, d_off, 0u, (size_t)n_qual, rocprim::plus<unsigned int>(), s));
        if (!(tmp = tmp_for(need))) return fail(c, PIE_E_NOMEM, "scan scratch of %zu bytes", need);
        PIE_HIP(c, rocprim::exclusive_scan(tmp, need, d_size, d_off, 0u, (size_t)n_qual, rocprim::plus<unsigned int>(), s));
        const unsigned gb = n_qual < (unsigned)c->n_cus * 16 ? n_qual : (unsigned)c->n_cus * 16;
        hipLaunchKernelGGL(k_arch_gather, dim3(gb ? gb : 1u), dim3(256), 0, s, d_val[1], d_nqual, d_ghead, d_size, d_off, v_out, d_queue);
        PIE_HIP(c, hipGetLastError());
        if (ev1) PIE_HIP(c, hipEventRecord(ev1, s));
        if (queue_out) PIE_HIP(c, hipMemcpyAsync(queue_out, d_queue, q * 4, hipMemcpyDeviceToHost, s));
        PIE_HIP(c, hipStreamSynchronize(s));
    } else if (ev1) {
        PIE_HIP(c, hipEventRecord(ev1, s));
        PIE_HIP(c, hipStreamSynchronize(s));
    }
    if (ev0) {
        float ms = 0;
        if (hipEventElapsedTime(&ms, ev0, ev1) == hipSuccess) { c->arch_ms_sum += ms; c->arch_calls++; }
        (void)hipEventDestroy(ev0);
        (void)hipEventDestroy(ev1);
    }
    return PIE_OK;
}

int pie_archive_stats(pie_ctx* c, double* ms_sum_out, uint32_t* calls_out, uint64_t* alg_bytes_out)
{
    if (!c) return PIE_E_INVAL;
    if (ms_sum_out) *ms_sum_out = c->arch_ms_sum;
    if (calls_out) *calls_out = c->arch_calls;
    if (alg_bytes_out) *alg_bytes_out = c->arch_alg_bytes;
    return PIE_OK;
}

int pie_set_scan_form(pie_ctx* c, int form)
{
    if (!c) return PIE_E_INVAL;
    if (c->n_flight || c->b_flight) return fail(c, PIE_E_STATE, "a scan is in flight");
    if (form < 0) {
        c->k1_pinned = false;
        c->k1_variant = 0x03;
    } else {
        c->k1_pinned = true;
        c->k1_variant = form;
    }
    return PIE_OK;
}

int pie_table_info_get(pie_ctx* c, pie_table_info* out)
{
    if (!c || !out) return PIE_E_INVAL;
    if (out->struct_size != sizeof(pie_table_info)) return fail(c, PIE_E_INVAL, "pie_table_info.struct_size mismatch");
    const size_t rows = (size_t)c->cap_rows, users = (size_t)c->cap_users;
    out->has_keys = (c->key_ok && c->d_key && c->d_pay && c->d_fkey) ? 1u : 0u;
    out->rows = (uint64_t)c->n;
    out->users = (uint64_t)c->n_users;
    out->table_bytes = (uint64_t)rows * 24u;
    out->derived_bytes = (c->d_key && c->d_pay && c->d_fkey) ? (uint64_t)rows * (sizeof(lkey_t) + sizeof(fkey_t) + sizeof(PayRec)) + 128u : 0u;
    // mirrors ensure_capacity(): per slot sel + sel_rank + bkt + out_idx (row-sized), the direct slots, the per-user arrays
    uint64_t per_slot = 0;
    if (rows) {
        per_slot += (uint64_t)c->sel_cap * (sizeof(SelRec) + 4) + (uint64_t)rows * (sizeof(BktRec) + 4);
        per_slot += (uint64_t)users * (8 + 4 + 4 + sizeof(Segment) * 2 + 4) + (uint64_t)(rows / kSegMax) * sizeof(Segment);
        if (c->slot[0].direct) per_slot += ((uint64_t)users << c->dshift) * sizeof(BktRec);
        per_slot += (uint64_t)kPartMax * kPartCap * sizeof(SelRec);
    }
    out->workspace_bytes = rows ? 2 * per_slot + 3 * (uint64_t)counts_span(c) : 0;
    for (bool alloc : c->lane_alloc) // batched scans, per lane: three slots (union bucket slots 20 B each, the union result 12 B per entry, 8 per user), three spans
        if (alloc)
            out->workspace_bytes += kBatchSlots * (((uint64_t)users << c->bdshift) * (sizeof(BktRec) + 4) + (uint64_t)batch_ucap(c) * 12 + ((uint64_t)users + 2) * 8) +
                                    3 * (uint64_t)batch_span_bytes(c);
    for (const BatchSlot& b : c->bslot) // per-query list storage, where somebody asked for lists
        out->workspace_bytes += (uint64_t)b.lists_q * ((uint64_t)batch_users_stride(c) * 12 + (uint64_t)batch_out_stride(c) * 4);
    out->index_build_ms = c->index_build_ms;
    out->ordered_rows = c->ord.valid ? (uint64_t)c->ord.held : 0u;
    out->ordered_positions = c->ord.valid ? (uint64_t)c->ord.n : 0u;
    // the run's columns (record 16 + end 8 + keys 3 per position, row -> position 4 per row) and its small per-unit arrays
    out->ordered_bytes = c->ord.pay ? (uint64_t)c->ord.pos_cap * (sizeof(OrdRec) + 8 + sizeof(lkey_t) + sizeof(fkey_t)) + (uint64_t)c->ord.cap * 4 +
                                          (uint64_t)c->ord.units_cap * 12 + ((uint64_t)c->ord.pos_cap / kOrdTile + 2) * kOrdSlices * 12 +
                                          ((uint64_t)c->ord.cap_users + 1) * 12
                                    : 0u;
    out->ordered_build_ms = c->ord.build_ms;
    out->ordered_builds = c->ord.builds;
    out->ordered_respreads = c->ord.respreads;
    return PIE_OK;
}

int pie_set_ordered_run(pie_ctx* c, int mode)
{
    if (!c) return PIE_E_INVAL;
    if (mode < 0 || mode > 2) return fail(c, PIE_E_INVAL, "ordered-run mode must be 0 (never), 1 (adaptive) or 2 (always)");
    if (c->n_flight || c->b_flight) return fail(c, PIE_E_STATE, "a scan is in flight");
    c->ord.mode = mode;
    c->ord.wanted = 0;
    if (mode == 0) {
        PIE_HIP(c, hipSetDevice(c->device));
        PIE_HIP(c, hipStreamSynchronize(c->stream));
        ord_free(c);
    }
    return PIE_OK;
}

int pie_host_alloc(pie_ctx* c, size_t bytes, void** host_out, void** dev_out)
{
    if (!c || !host_out || !dev_out || bytes == 0) return PIE_E_INVAL;
    PIE_HIP(c, hipSetDevice(c->device));
    void* h = nullptr;
    void* d = nullptr;
    PIE_HIP(c, hipHostMalloc(&h, bytes, hipHostMallocMapped));
    hipError_t e = hipHostGetDevicePointer(&d, h, 0);
    if (e != hipSuccess) { (void)hipHostFree(h); return fail(c, PIE_E_HIP, "hipHostGetDevicePointer: %s", hipGetErrorString(e)); }
    memset(h, 0, bytes);
    *host_out = h;
    *dev_out = d;
    return PIE_OK;
}

int pie_host_free(pie_ctx* c, void* host_ptr)
{
    if (!c) return PIE_E_INVAL;
    if (!host_ptr) return PIE_OK;
    PIE_HIP(c, hipSetDevice(c->device));
    PIE_HIP(c, hipStreamSynchronize(c->stream)); // a scan may still be writing it
    PIE_HIP(c, hipHostFree(host_ptr));
    return PIE_OK;
}

int pie_set_profiling(pie_ctx* c, int enabled)
{
    if (!c) return PIE_E_INVAL;
    c->profiling = enabled != 0;
    c->profile_every = enabled > 1 ? enabled : 1;
    return PIE_OK;
}

int pie_stats_get(pie_ctx* c, pie_stats* out)
{
    if (!c || !out) return PIE_E_INVAL;
    if (out->struct_size != sizeof(pie_stats)) return fail(c, PIE_E_INVAL, "pie_stats.struct_size mismatch");
    PIE_HIP(c, hipSetDevice(c->device));
    int rc = resolve_events(c);
    if (rc) return rc;
    const Slot* sl = c->res;
    out->n_profiled = c->n_profiled;
    out->rows = (uint64_t)c->n;
    out->users = (uint64_t)c->n_users;
    out->selected = sl ? sl->last.m : 0;
    out->alg_bytes = 24ull * (uint64_t)c->n;
    out->k1_ms_sum = c->k1_ms_sum;
    out->scan_ms_sum = c->scan_ms_sum;
    out->max_bucket = sl ? sl->last.max_count : 0;
    out->n_segments = sl ? sl->last.n_seg : 0;
    out->n_big = sl ? sl->last.n_big : 0;
    out->k1_blocks = sl ? (uint32_t)sl->k1_blocks : 0;
    out->k1_variant = sl ? (uint32_t)sl->variant : 0;
    out->key_ambiguous = (c->res && (c->res->variant & 0x400)) ? (uint32_t)(c->res->last.amb > 0xFFFFFFFFull ? 0xFFFFFFFFull : c->res->last.amb) : 0u;
    out->live = sl ? sl->last.live : 0;
    out->candidates = (sl && (sl->variant & 0x400)) ? sl->last.cand : 0;
    if (c->last_was_batch && c->bres && c->bres->have_result) {
        // the last finished thing was a batch: selected = rows over all its queries, form = keyed | 0x1000 (batched)
        const BatchSlot& b = *c->bres;
        uint64_t m = 0;
        uint32_t mx = 0;
        for (int q = 0; q < b.n_q; ++q) { m += b.last[q].m; mx = mx > b.last[q].max_count ? mx : b.last[q].max_count; }
        out->selected = m;
        out->max_bucket = mx;
        out->n_segments = out->n_big = 0;
        out->k1_blocks = (uint32_t)b.k1_blocks;
        out->k1_variant = b.unsupported ? 0u : b.ordered ? (0x3400u | (b.fine_key ? 0x800u : 0u)) : (0x1485u | (b.fine_key ? 0x800u : 0u));
        out->key_ambiguous = 0;
        out->live = 0;
        out->candidates = b.unsupported ? 0 : b.last[0].cand;
    }
    return PIE_OK;
}

int pie_stats_reset(pie_ctx* c)
{
    if (!c) return PIE_E_INVAL;
    int rc = resolve_events(c);
    c->k1_ms_sum = c->scan_ms_sum = 0;
    c->arch_ms_sum = 0;
    c->arch_calls = 0;
    c->n_profiled = 0;
    return rc;
}

int pie_synchronize(pie_ctx* c)
{
    if (!c) return PIE_E_INVAL;
    PIE_HIP(c, hipSetDevice(c->device));
    return sync_all(c);
}

int pie_shard_table(pie_ctx* c, int32_t rank, int32_t world, size_t* n_rows_out, int32_t* n_users_out)
{
    if (!c) return PIE_E_INVAL;
    if (world < 1 || rank < 0 || rank >= world) return fail(c, PIE_E_INVAL, "rank %d outside [0, %d)", rank, world);
    if (c->cap_rows == 0) return fail(c, PIE_E_STATE, "no table loaded");
    if (c->n_flight || c->b_flight) return fail(c, PIE_E_STATE, "a scan is in flight");
    PIE_HIP(c, hipSetDevice(c->device));
    int rc = sync_all(c);
    if (rc) return rc;
    hipStream_t s = c->stream;
    const long long n = c->n;
    const int U = c->n_users;
    int *d_flag = nullptr, *d_local = nullptr, *d_users = nullptr, *d_blk = nullptr, *o_user = nullptr, *o_disc = nullptr, *o_row = nullptr;
    long long *d_uoff = nullptr, *d_boff = nullptr, *o_start = nullptr, *o_end = nullptr;
    auto cleanup = [&]() {
        dfree(d_flag); dfree(d_local); dfree(d_blk); dfree(d_uoff); dfree(d_boff);
        dfree(o_start); dfree(o_end); dfree(o_user); dfree(o_disc);
    };
#define PIE_TRY(call)                                                                                 \
    do {                                                                                              \
        hipError_t e_ = (call);                                                                       \
        if (e_ != hipSuccess) {                                                                       \
            cleanup();                                                                                \
            dfree(d_users);                                                                           \
            dfree(o_row);                                                                             \
            return fail(c, e_ == hipErrorOutOfMemory ? PIE_E_NOMEM : PIE_E_HIP, "%s: %s", #call, hipGetErrorString(e_)); \
        }                                                                                             \
    } while (0)
    // 1. which users stay, and their local ids
    const unsigned ub = (unsigned)((U + 255) / 256);
    PIE_TRY(hipMalloc(&d_flag, (size_t)U * 4));
    PIE_TRY(hipMalloc(&d_local, (size_t)U * 4));
    PIE_TRY(hipMalloc(&d_users, (size_t)U * 4));
    PIE_TRY(hipMalloc(&d_uoff, ((size_t)U + 1) * 8));
    hipLaunchKernelGGL(k_shard_user_flags, dim3(ub), dim3(256), 0, s, U, (int)rank, (int)world, d_flag);
    hipLaunchKernelGGL(k_block_prefix, dim3(1), dim3(256), 0, s, d_flag, U, d_uoff, (unsigned long long*)nullptr);
    hipLaunchKernelGGL(k_shard_user_ids, dim3(ub), dim3(256), 0, s, U, d_flag, d_uoff, d_local, d_users);
    PIE_TRY(hipGetLastError());
    long long u_local = 0;
    PIE_TRY(hipMemcpyAsync(&u_local, d_uoff + U, 8, hipMemcpyDeviceToHost, s));
    // 2. rows: per-block counts -> prefix -> order-preserving compaction into fresh columns
    const long long rpb = 256LL * 64;
    const int blocks = (int)((n + rpb - 1) / rpb) > 0 ? (int)((n + rpb - 1) / rpb) : 1;
    PIE_TRY(hipMalloc(&d_blk, (size_t)blocks * 4));
    PIE_TRY(hipMalloc(&d_boff, ((size_t)blocks + 1) * 8));
    hipLaunchKernelGGL(k_shard_row_count, dim3(blocks), dim3(256), 0, s, c->d_user, n, rpb, d_local, U, d_blk);
    hipLaunchKernelGGL(k_block_prefix, dim3(1), dim3(256), 0, s, d_blk, blocks, d_boff, (unsigned long long*)nullptr);
    PIE_TRY(hipGetLastError());
    long long n_local = 0;
    PIE_TRY(hipMemcpyAsync(&n_local, d_boff + blocks, 8, hipMemcpyDeviceToHost, s));
    PIE_TRY(hipStreamSynchronize(s));
    const size_t rows = (size_t)(n_local > 0 ? n_local : 1);
    PIE_TRY(hipMalloc(&o_start, rows * 8));
    PIE_TRY(hipMalloc(&o_end, rows * 8));
    PIE_TRY(hipMalloc(&o_user, rows * 4));
    PIE_TRY(hipMalloc(&o_disc, rows * 4));
    PIE_TRY(hipMalloc(&o_row, rows * 4));
    if (n > 0) {
        hipLaunchKernelGGL(k_shard_row_write, dim3(blocks), dim3(256), 0, s, c->d_start, c->d_end, c->d_user, c->d_disc, n, rpb, d_local, U,
                           d_boff, o_start, o_end, o_user, o_disc, o_row);
        PIE_TRY(hipGetLastError());
        PIE_TRY(hipStreamSynchronize(s));
    }
    // 3. a right-sized table takes the shard (the whole-table buffers and their workspace are released first)
    free_table(c);
    const int users_new = u_local > 0 ? (int)u_local : 1;
    rc = ensure_capacity(c, n_local, users_new);
    if (rc) { cleanup(); dfree(d_users); dfree(o_row); return rc; }
    if (n_local > 0) {
        PIE_TRY(hipMemcpyAsync(c->d_start, o_start, (size_t)n_local * 8, hipMemcpyDeviceToDevice, s));
        PIE_TRY(hipMemcpyAsync(c->d_end, o_end, (size_t)n_local * 8, hipMemcpyDeviceToDevice, s));
        PIE_TRY(hipMemcpyAsync(c->d_user, o_user, (size_t)n_local * 4, hipMemcpyDeviceToDevice, s));
        PIE_TRY(hipMemcpyAsync(c->d_disc, o_disc, (size_t)n_local * 4, hipMemcpyDeviceToDevice, s));
        PIE_TRY(hipStreamSynchronize(s));
    }
#undef PIE_TRY
    cleanup();
    dfree(c->d_shard_rows);
    dfree(c->d_shard_users);
    c->d_shard_rows = o_row;
    c->d_shard_users = d_users;
    c->shard_rows_n = n_local;
    c->shard_users_n = (int)u_local;
    if (n_rows_out) *n_rows_out = (size_t)n_local;
    if (n_users_out) *n_users_out = users_new;
    rc = build_keys(c, 0);
    if (rc) return rc;
    PIE_HIP(c, hipStreamSynchronize(s));
    return PIE_OK;
}

int pie_shard_maps(pie_ctx* c, int32_t* rows_global_out, int32_t* users_global_out)
{
    if (!c) return PIE_E_INVAL;
    if (!c->d_shard_rows || !c->d_shard_users) return fail(c, PIE_E_STATE, "pie_shard_maps without pie_shard_table");
    PIE_HIP(c, hipSetDevice(c->device));
    if (rows_global_out && c->shard_rows_n)
        PIE_HIP(c, hipMemcpyAsync(rows_global_out, c->d_shard_rows, (size_t)c->shard_rows_n * 4, hipMemcpyDeviceToHost, c->stream));
    if (users_global_out && c->shard_users_n)
        PIE_HIP(c, hipMemcpyAsync(users_global_out, c->d_shard_users, (size_t)c->shard_users_n * 4, hipMemcpyDeviceToHost, c->stream));
    PIE_HIP(c, hipStreamSynchronize(c->stream));
    return PIE_OK;
}

int32_t pie_shard_of(int32_t user, int32_t n_shards)
{
    if (n_shards <= 1) return 0;
    unsigned long long z = (unsigned long long)(uint32_t)user + 0x9E3779B97F4A7C15ULL;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    z ^= z >> 31;
    return (int32_t)(z % (unsigned long long)n_shards);
}

} // extern "C"
