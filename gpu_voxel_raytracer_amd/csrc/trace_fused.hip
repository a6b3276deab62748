// trace_fused.hip — fused_kernel: head and compacted tail of a trace launch in ONE grid of persistent waves (VXRT_OPT_FUSED_TAIL).
// -DVXRT_VARIANTS=1 builds only (VARIANT_SOURCES): measured slower than trace_kernel + bounce_kernel, kept as the record of the attempt
// with its parity cases (HISTORY.md section 11).  Moved out of trace.hip in round 6.
#include "trace_block.h"

namespace vxrt {
namespace {

// ---- fused_kernel: head and compacted tail of a launch in ONE grid of persistent waves (VXRT_OPT_FUSED_TAIL; -DVXRT_VARIANTS=1 only) --------
// MEASURED SLOWER (round 5; HISTORY.md section 11): bit-identical images, 0.50 ms against 0.41 (two kernels) and 0.375 (all-in-one) for a
// rank of 8's 20-frame block.  A software scheduler pays for every decision with round trips through device-scope memory (2-5 us each:
// a claim, a look at the shard counters, a flag) where the hardware's dispatcher starts the next wave for nothing, and it holds both
// kernels' bodies (4 waves per SIMD) in waves that sleep in their slots when they have nothing to do.  Kept beside tracers 2, 3 and 5
// as the record of the attempt, with its parity cases.
// A launch that is little more than its longest chains — a rank's share of a short block on many GPUs: 20 frames of an eighth of the
// rows — spends its time DRAINING: trace_kernel ends when its longest wave ends (the chip two thirds idle by then), and only then may
// bounce_kernel start, which drains again.  Two chains end to end, at 83 % and 64 % of the instruction rate the same kernels reach in
// the steady state (round 5: profiles/r05/short_block_timelines.txt).  Here the waves are persistent: each takes the launch's blocks
// (tile x frame group, in the launch order: longest first) from a cursor, and when those run out it takes CHUNKS of 64 queued paths,
// as soon as a chunk is complete — the tail of the paths handed over early runs beside the long head chains, and nothing waits for a
// kernel boundary.  Same records, same per-path operations (trace_block, bounce_path): same image.
//
// Hot words.  One memory channel takes ~90 atomics (or agent-scope loads) per microsecond — the ray counters taught that in round 1, and
// the first version of this kernel, with one cursor block of 1 KB and every idle wave polling the same three words, spent 76 % of its
// wave-cycles in s_waitcnt (profiles/r05/fused_kernel_counters.txt).  So: kCursors head cursors 256 bytes apart, cursor k over the
// blocks k, k + kCursors, ... (the launch order survives in each); a wave starts at its own and goes round when that is dry; ONE block
// per claim among the tiles that walk (the front of the order), four among the tiles of sky.  A wave that leaves the head phase adds
// its blocks to one counter (one atomic per wave); the wave that completes the total raises kFlags copies of a flag, and an idle
// wave looks only at its own copy, at the four shard counters of its own group and at their four chunk cursors, every ~14 us.
//
// Hand-over.  A head lane reserves its record slot with the shard's counter as before and stores the record with queue_store_fused:
// seven 8-byte stores at agent scope (write-through, visible to every XCD), s_waitcnt, then the eighth, which holds the launch's
// 16-bit stamp in the unused top of normal_ambient.  A consumer lane polls that word (agent scope) until the stamp is there, reads
// the rest and writes the word back with stamp 0.  Before the heads are done only chunks whose 64 slots are all reserved are taken;
// afterwards every shard counter is final and the part-filled last chunks go too.
//
// Waiting is bounded in three ways: a wave only waits for work that resident, running waves are producing (a block is claimed by a
// wave that is already running, so no wave ever waits for one that has no slot); every spin sleeps; a spin that lasts longer than
// any legitimate wait (tens of milliseconds) sets ctl->error and gives up — the frame is then wrong and vxrt_sync reports it, but
// the grid drains.
constexpr unsigned kCursors = 64, kFlags = 64, kHotStride = 64;   // uints: 256 bytes between hot words
struct FusedCtl {
    unsigned cursor[kCursors * kHotStride];        // head work cursors (local block index of cursor k)
    unsigned done_flag[kFlags * kHotStride];       // copies of "every head block is finished"
    unsigned heavy_flag[kFlags * kHotStride];      // copies of "every block of a tile that walked last time is finished": part-filled chunks may be closed
    unsigned blocks_done[kHotStride];              // head blocks finished (their records stored and stamped), added wave by wave
    unsigned heavy_done[kHotStride];               // ... of those, blocks before heavy_blocks
    unsigned error[kHotStride];                    // a bounded wait ran out
    unsigned next_chunk[kShards * kCountStride];   // per shard: the next chunk nobody has claimed yet
    // diagnostics (vxrt_debug_fused_profile): shader clock ~earliest wave start (stored inverted), the clock when the done flags went
    // up, the latest wave end; chunks taken before / after the flags; idle sleeps; stamp polls; head claims
    unsigned long long prof[8];
};

__device__ __forceinline__ unsigned agent_load(const unsigned* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
// ... and one that has RETURNED before anything after it is issued: two loads sent off back to back may be sampled in either order, and
// "the flag is up, so the counter I read is final" needs the flag sampled first
__device__ __forceinline__ unsigned agent_load_first(const unsigned* p) {
    const unsigned v = __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    return v;
}

#ifndef VXRT_FUSED_WAVES
#define VXRT_FUSED_WAVES 4   // waves per SIMD of fused_kernel: 128 VGPRs — at 5 (96) the persistent state spills into the walk's loops
#endif
template <int kF>
__global__ __launch_bounds__(kTB, VXRT_FUSED_WAVES) void fused_kernel(const TraceArgs a, FusedCtl* ctl, const unsigned total_blocks, const uint32_t* sort_info,
                                                                      const uint32_t stamp, const int first_bounce) {
    static_assert(kTB == 64 && kCursors == 64, "one wave per block; lane c reads cursor c");
    extern __shared__ uint4 lds_stack[];
    const int lane = threadIdx.x & 63;
    zero_counts(a.tail_zero, threadIdx.x);
    const bool prof_wave = lane == 0 && blockIdx.x % 64u == 0u;    // every 64th wave reports (the counter line takes ~90 atomics per us)
    if (prof_wave) atomicMax(&ctl->prof[0], ~(unsigned long long)__builtin_amdgcn_s_memrealtime());
    unsigned prof_claims = 0, prof_idle = 0, prof_polls = 0, prof_before = 0, prof_after = 0;
    // where the tiles that walk end in the launch order (tile_scan_kernel left the count and the spread beside the sort's histogram;
    // spread_position puts them at the front, each followed by k tiles of sky): unknown (the stream's first launch) -> 0
    unsigned heavy_blocks = 0;
    if (sort_info != nullptr && a.tile_order != nullptr) {
        const unsigned tiles = total_blocks / unsigned(a.batch), nh = sort_info[0], nl = tiles - (nh < tiles ? nh : tiles);
        const unsigned k = nh ? unsigned((unsigned long long)nl * sort_info[1] / 256u / nh) & ~1u : 0u;
        const unsigned long long hb = (unsigned long long)(nh < tiles ? nh : tiles) * (k + 1u) * unsigned(a.batch);
        heavy_blocks = hb > total_blocks ? total_blocks : unsigned(hb);
    }
    // ---- one loop, three kinds of work, in this order of preference:
    //   1. a block of a tile that WALKS (the front of the launch order): the launch's critical path — claimed one at a time, run at
    //      the highest priority;
    //   2. a chunk of 64 queued paths that is complete (its tail is the second half of the critical path);
    //   3. blocks of SKY, four per claim: filler, 5 us each, which nothing waits for.
    // (The first version ran all of 1 and 3 before any of 2: the blocks of sky at the end of the order kept every wave in the head
    // phase until it was over, and the "fused" launch was two phases again: 324 us to the last head block, 230 us of tail behind it.)
    const Caster<false> caster(a, lds_stack, int(threadIdx.x));
    const f3 sun_dir = ld3(a.sun_dir), sun_color = ld3(a.sun_color), sky = ld3(a.sky_color);
    const PathQueue none{nullptr, nullptr, 0u};
    uint32_t rays = 0;
    const unsigned group = blockIdx.x % 16u;       // before part-filled chunks may be taken this wave serves the shards group, group + 16, + 32, + 48
    const unsigned home = blockIdx.x % kShards;
    const unsigned* my_flag = &ctl->done_flag[(blockIdx.x % kFlags) * kHotStride];
    const unsigned* my_heavy_flag = &ctl->heavy_flag[(blockIdx.x % kFlags) * kHotStride];
    unsigned my_blocks = 0, my_heavy = 0, idle = 0;
    unsigned k = blockIdx.x % kCursors, last = 0;  // the head cursor this wave claims from, and where it stood at the wave's last claim
    bool head_dry = total_blocks == 0u;
    auto report_heavy = [&]() {   // this wave's finished blocks of walking tiles -> the counter; the wave that completes it raises the flags
        if (my_heavy == 0u) return;
        __builtin_amdgcn_s_waitcnt(0);             // their records are written through
        if (lane == 0) {
            const unsigned before = atomicAdd(&ctl->heavy_done[0], my_heavy);
            if (before + my_heavy >= heavy_blocks)
                for (unsigned f = 0; f < kFlags; f++) __hip_atomic_store(&ctl->heavy_flag[f * kHotStride], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        my_heavy = 0;
    };
    auto report_blocks = [&]() {  // ... and all its finished blocks, once, when the cursors are dry
        __builtin_amdgcn_s_waitcnt(0);
        if (lane == 0 && my_blocks != 0u) {
            const unsigned before = atomicAdd(&ctl->blocks_done[0], my_blocks);
            if (before + my_blocks >= total_blocks) {  // the last head wave: every record is stored, every shard counter final
                for (unsigned f = 0; f < kFlags; f++) __hip_atomic_store(&ctl->done_flag[f * kHotStride], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                ctl->prof[1] = __builtin_amdgcn_s_memrealtime();
            }
        }
        my_blocks = 0;
    };
    // claims `step` blocks of cursor k; false: k is dry — then k moves to a cursor that is not, or head_dry is set
    unsigned first_j = 0, count_j = 0;
    // what was asked for before the last blocks of sky ran: a claim of four more (lane 0 holds the answer), and a look at the chunks
    bool sky_pending = false, pf = false, pf_mine = false, closing_seen = false;
    // a chunk that was claimed part-filled while a claim of sky is in hand: waiting for it may mean waiting for the done flag, i.e. for
    // this wave's own unrun blocks — so it is set aside until they have run
    bool held = false;
    unsigned held_shard = 0, held_chunk = 0, held_ns = 0;
    unsigned pf_sky_i = 0, pf_k = 0, pf_done = 0, pf_hv = 0, pf_n = 0, pf_next = 0;
    auto claim_head = [&](unsigned step) -> bool {
        for (;;) {
            const unsigned per_cursor = total_blocks > k ? (total_blocks - k + kCursors - 1u) / kCursors : 0u;   // blocks k, k + kCursors, ... < total_blocks
            unsigned i = 0;
            if (lane == 0) i = atomicAdd(&ctl->cursor[k * kHotStride], step);
            i = __builtin_amdgcn_readfirstlane(i);
            prof_claims++;
            if (i < per_cursor) { first_j = i; count_j = i + step <= per_cursor ? step : per_cursor - i; last = i + step; return true; }
            // this cursor is dry for good.  ONE look at all of them (lane c reads cursor c) instead of 63 more failed claims — which is
            // what every wave did at first, 330 000 atomics per launch on 64 lines — then on to the first after this one that is not
            const unsigned c = unsigned(lane) % kCursors;
            const unsigned per_c = total_blocks > c ? (total_blocks - c + kCursors - 1u) / kCursors : 0u;
            unsigned long long m = __ballot(agent_load(&ctl->cursor[c * kHotStride]) < per_c);
            if (m == 0ull) { head_dry = true; return false; }                                // every block is taken
            const unsigned r = (k + 1u) % kCursors;
            m = r == 0u ? m : (m >> r | m << (64u - r));
            k = (r + unsigned(__ffsll((long long)m) - 1)) % kCursors;
            last = 0xffffffffu / kCursors;                                                   // its front is gone by now
            step = 4u;
        }
    };
    for (;;) {
        // ---- what next?
        bool do_head = false, do_chunk = false;
        unsigned shard = 0, chunk = 0, n_s = 0;
        bool heads_done = false;
        if (head_dry && my_blocks != 0u && !sky_pending) { report_heavy(); report_blocks(); }   // the cursors are dry: this wave's blocks count now
        if (!head_dry && !sky_pending && last * kCursors + k < heavy_blocks) do_head = claim_head(1u);   // 1. a walking tile
        // 2. a chunk — from the look that was sent off BEFORE the last blocks of sky ran (pf), or from a look made now.  (With a claim
        // of sky in hand and no look, the blocks go first: they send the next look off.)
        if (!do_head && held && !sky_pending) { do_chunk = true; shard = held_shard; chunk = held_chunk; n_s = held_ns; held = false; pf = false; }
        if (!do_head && !do_chunk && !held && (pf || !sky_pending)) {
            const unsigned q = unsigned(lane);      // lane q looks at shard q — before `closing` only the four lanes of this wave's group do
            // The look that travelled while the blocks ran is good for ONE thing: "a whole chunk is reserved" (a counter only grows,
            // and the claim below is checked against it).  Its flags and its counters were sampled in no particular order, so
            // whatever depends on the flags — taking a part-filled chunk, deciding that nothing is left — is decided by a look made
            // now, flag first.
            for (bool fresh = !pf;; fresh = true) {
                unsigned n = 0, next = 0;
                bool mine, closing = false;
                if (!fresh) {
                    mine = pf_mine; n = pf_n; next = pf_next;
                    pf = false;
                } else {
                    heads_done = agent_load_first(my_flag) != 0u;                            // sampled BEFORE the counters: then they are final
                    // once the tiles that walked last time are through, a part-filled chunk is worth taking: whoever takes it closes it
                    closing = closing_seen = heads_done || (heavy_blocks != 0u && agent_load(my_heavy_flag) != 0u);
                    mine = closing || (q % 16u) == group;
                    if (mine) {
                        n = agent_load(a.tail.counts + q * kCountStride);
                        next = agent_load(&ctl->next_chunk[q * kCountStride]);
                    }
                }
                n = n < a.tail.shard_capacity ? n : a.tail.shard_capacity;
                const bool open = mine && ((next + 1u) * 64u <= n || (closing && next * 64u < n));
                unsigned long long m = __ballot(open);
                m = home == 0u ? m : (m >> home | m << (64u - home));                       // rotate: bit 0 = the wave's own shard
                if (m != 0ull) {
                    shard = (home + unsigned(__ffsll((long long)m) - 1)) % kShards;
                    n_s = __shfl(n, int(shard), 64);
                    if (lane == 0) chunk = atomicAdd(&ctl->next_chunk[shard * kCountStride], 1u);
                    chunk = __builtin_amdgcn_readfirstlane(chunk);
                    do_chunk = !(heads_done && chunk * 64u >= n_s);                          // (somebody else took the shard's last chunk)
                    break;
                }
                if (fresh || (pf_done | pf_hv) == 0u) break;                                 // nothing; or the old look's flags say: look properly
            }
        }
        if (!do_head && !do_chunk && sky_pending) {                                          // 3. sky: the claim sent off before the last blocks ran
            sky_pending = false;
            const unsigned i = __builtin_amdgcn_readfirstlane(pf_sky_i);
            const unsigned per_cursor = total_blocks > pf_k ? (total_blocks - pf_k + kCursors - 1u) / kCursors : 0u;
            if (i < per_cursor) { k = pf_k; first_j = i; count_j = i + 4u <= per_cursor ? 4u : per_cursor - i; last = i + 4u; do_head = true; }
        }
        if (!do_head && !do_chunk && !head_dry) do_head = claim_head(4u);                    //    ... or a claim made now
        if (!do_head && !do_chunk) {
            if (heads_done) break;                 // every counter is final and every chunk is claimed: done
            idle++;
            prof_idle++;
            for (unsigned z = 0; z < 4u; z++) __builtin_amdgcn_s_sleep(127);               // ~14 us
            if (idle > (1u << 13)) { if (lane == 0) atomicOr(&ctl->error[0], 1u); break; }  // > 0.1 s of nothing: give up
            continue;
        }
        idle = 0;
        if (do_head) {
            if (first_j * kCursors + k >= heavy_blocks && !head_dry && !held) {
                // Blocks of sky: what the wave will want to know when they are done is asked for NOW — the next claim of four and a
                // look at the chunks — so that the answers travel while the blocks run.  (Every decision of this scheduler is a round
                // trip through device-scope memory, 2-5 us; made one after the other they cost more than the 20 us of sky between them.)
                // Never among the walking tiles: a wave that held two of the longest chains would run them one after the other.
                pf_k = k;
                pf_sky_i = 0;
                if (lane == 0) pf_sky_i = atomicAdd(&ctl->cursor[k * kHotStride], 4u);
                prof_claims++;
                sky_pending = true;
                pf_done = agent_load(my_flag);
                pf_hv = agent_load(my_heavy_flag);
                pf_mine = closing_seen || (unsigned(lane) % 16u) == group;
                pf_n = 0; pf_next = 0;
                if (pf_mine) {
                    pf_n = agent_load(a.tail.counts + unsigned(lane) * kCountStride);
                    pf_next = agent_load(&ctl->next_chunk[unsigned(lane) * kCountStride]);
                }
                pf = true;
            }
            for (unsigned j = first_j; j < first_j + count_j; j++) {
                const unsigned b = j * kCursors + k;
                // a walking tile's chain is the launch's critical path, and here it shares its SIMD with chunks and blocks of sky
                if (b < heavy_blocks) __builtin_amdgcn_s_setprio(3);
                trace_block<false, kF, true>(a, b, lds_stack, stamp);
                __builtin_amdgcn_s_setprio(0);
                my_blocks++;
                if (b < heavy_blocks) my_heavy++;
            }
            if (my_heavy != 0u && last * kCursors + k >= heavy_blocks) report_heavy();       // this wave has left the front of the order
            continue;
        }
        // ---- a chunk.  Which of its 64 slots hold (or will hold) a record: all of them if the chunk is full; for a part-filled chunk,
        // wait until it fills, or — once part-filled chunks may be taken — CLOSE it: the shard's counter jumps from n to the chunk's
        // end, so that later records start the next chunk, and the slots from n on stay empty.  Wave-uniform.
        __builtin_amdgcn_s_setprio(1);             // a chunk is a chain too: ahead of the blocks of sky, behind the walking tiles
        const unsigned chunk_end = (chunk + 1u) * 64u < a.tail.shard_capacity ? (chunk + 1u) * 64u : a.tail.shard_capacity;
        unsigned limit = chunk_end;
        if (n_s < chunk_end && sky_pending) {      // (see `held`)
            held = true; held_shard = shard; held_chunk = chunk; held_ns = n_s;
            __builtin_amdgcn_s_setprio(0);
            continue;
        }
        if (n_s < chunk_end) {
            report_heavy();                        // this wait may depend on the flags: nothing this wave has finished may be missing from them
            report_blocks();
            for (unsigned tries = 0;; tries++) {
                const bool done_now = agent_load_first(my_flag) != 0u;                      // sampled before the counter: then it is final
                const unsigned cnt = agent_load(a.tail.counts + shard * kCountStride);
                if (cnt >= chunk_end) break;                                                // filled meanwhile
                if (done_now) { limit = cnt > chunk * 64u ? cnt : chunk * 64u; break; }     // final: what is there is all there will be
                // (only the chunk at the shard's fill front is closed — cnt > chunk * 64: a jump over an earlier chunk's free slots would
                // make its owner wait for records that never come)
                if (heavy_blocks != 0u && cnt > chunk * 64u && agent_load(my_heavy_flag) != 0u && chunk_end == (chunk + 1u) * 64u) {
                    unsigned old = 0;
                    if (lane == 0) old = atomicCAS(a.tail.counts + shard * kCountStride, cnt, chunk_end);
                    old = __builtin_amdgcn_readfirstlane(old);
                    if (old == cnt) { limit = cnt; break; }                                  // closed with cnt records
                    continue;                                                               // the counter moved: look again
                }
                __builtin_amdgcn_s_sleep(100);
                if (tries > (1u << 15)) { if (lane == 0) atomicOr(&ctl->error[0], 8u); limit = chunk * 64u; break; }
            }
        }
        if (heads_done) prof_after++; else prof_before++;
        const unsigned entry = chunk * 64u + unsigned(lane);
        const bool in_queue = entry < limit;
        float4* slot = a.tail.recs + (size_t(shard) * a.tail.shard_capacity + (in_queue ? entry : 0u)) * 4u;
        unsigned long long* word = reinterpret_cast<unsigned long long*>(slot) + 3;          // dir.z | normal_ambient, stamp on top
        bool valid = false;
        unsigned long long w3 = 0ull;
        if (in_queue) {   // reserved: the record is there or on its way
            for (unsigned polls = 0;; polls++) {
                w3 = __hip_atomic_load(word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (unsigned(w3 >> 48) == stamp) { valid = true; break; }
                if (lane == 0) prof_polls++;
                if (polls < 16u) __builtin_amdgcn_s_sleep(4); else __builtin_amdgcn_s_sleep(64);
                if (polls > (1u << 14)) { atomicOr(&ctl->error[0], 2u); break; }
            }
        }
        if (valid) {
            const PathRec rec = load_rec_fused(slot, w3);
            __hip_atomic_store(word, w3 & 0x0000ffffffffffffull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // stamp 0: the slot is free for the next launch
            bounce_path<false>(a, caster, rec, none, shard, first_bounce, a.max_bounces, sun_dir, sun_color, sky, rays);
        }
        __builtin_amdgcn_s_setprio(0);
    }
    count_rays(a.ray_counter, rays, lane);
    if (prof_wave) {
        atomicMax(&ctl->prof[2], (unsigned long long)__builtin_amdgcn_s_memrealtime());
        atomicAdd(&ctl->prof[3], (unsigned long long)prof_before);
        atomicAdd(&ctl->prof[4], (unsigned long long)prof_after);
        atomicAdd(&ctl->prof[5], (unsigned long long)prof_idle);
        atomicAdd(&ctl->prof[6], (unsigned long long)prof_polls);
        atomicAdd(&ctl->prof[7], (unsigned long long)prof_claims);
    }
}


}  // namespace

size_t fused_ctl_bytes() { return sizeof(FusedCtl); }

// One grid of persistent one-wave blocks for the whole launch (fused_kernel).  ctl: device memory of fused_ctl_bytes(), zeroed on the
// stream before this call.  sort_scratch: the tile sort's scratch of the order in a.tile_order (null: none yet — every block is claimed in fours).
// stamp: 1 .. 65535, different from the previous launch's on the same queue.
hipError_t launch_fused(const TraceArgs& args, void* ctl, unsigned waves, const uint32_t* sort_scratch, uint32_t stamp, hipStream_t s) {
    TraceArgs a = args;
    a.block_first = 0;
    if (kTB != 64) return hipErrorInvalidValue;
    const unsigned all_blocks = trace_tile_count(a.band.width, a.band.local_rows) * unsigned(a.batch);
    const uint32_t* sort_info = sort_scratch ? sort_scratch + kSortBins * kSortBlocks : nullptr;
    const size_t lds = caster_lds_bytes(a, false, kTB);
    dim3 grid(waves < 1u ? 1u : waves);
    FusedCtl* fc = static_cast<FusedCtl*>(ctl);
    constexpr int kF8 = kTB == 64 ? 8 : 1, kF4 = kTB == 64 ? 4 : 1;
    if (a.frame_lanes == 8) hipLaunchKernelGGL((fused_kernel<kF8>), grid, dim3(kTB), lds, s, a, fc, all_blocks, sort_info, stamp, a.tail_from);
    else if (a.frame_lanes == 4) hipLaunchKernelGGL((fused_kernel<kF4>), grid, dim3(kTB), lds, s, a, fc, all_blocks, sort_info, stamp, a.tail_from);
    else hipLaunchKernelGGL((fused_kernel<1>), grid, dim3(kTB), lds, s, a, fc, all_blocks, sort_info, stamp, a.tail_from);
    return hipGetLastError();
}

// byte offset of the error word in the control block (the host copies the block's head back after a launch)
size_t fused_ctl_error_offset() { return offsetof(FusedCtl, error); }
size_t fused_ctl_profile_offset() { return offsetof(FusedCtl, prof); }


}  // namespace vxrt
