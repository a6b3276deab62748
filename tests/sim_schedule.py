#!/usr/bin/env python3
"""ORACLE-BASED DIAGNOSTIC (test infrastructure: lives under tests/ because it calls the oracle; not collected by pytest).
CPU-side what-if for the monolithic tracer's wave scheduling (no GPU): takes the oracle's per-ray octree step
counts of the bench frame (orc_trace_steps) and prices three ways of running an 8x8 tile on one wave64:

  sync   every lane casts its next ray, the wave leaves the walk loop when the LAST lane's ray ends, then all shade
         (trace.hip today);
  gated  lanes run free: a lane whose ray ended waits; the shading block runs when >= T lanes wait or nobody walks;
  dense  lower bound: all steps and all shading events packed 64 per instruction.

Costs are in wave-instructions: C_STEP per walk trip, C_SHADE per shading block execution (includes walk_begin).
"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))  # tests/ -> repo root
sys.path.insert(0, ROOT)
from gpu_voxel_raytracer_amd import scenes  # noqa: E402
from oracle import oracle as O  # noqa: E402

C_STEP = float(os.environ.get("C_STEP", 100))
C_SHADE = float(os.environ.get("C_SHADE", 450))


def steps_for(scene="menger", view="bench", w=1920, h=1080, bounces=4):
    pos, mrgb, size = scenes.load_scene(scene)
    cam = scenes.bench_camera(size) if view == "bench" else scenes.close_camera(size)
    octree = O.create_octree(pos, mrgb)
    noise = O.noise_table()
    u = O.Uniforms.default()
    u.set_camera(cam[0], O.camera_axis_scaled(cam[0], cam[1], cam[2], w, h))
    u.frame_number = 1
    out = np.zeros((h, w, 17), np.int32)
    O.lib().orc_trace_steps(O._p(octree), O._p(noise), C.byref(u), C.c_int(bounces), C.c_int(0), C.c_int(0), C.c_int(w),
                            C.c_int(h), O._p(out), C.c_int(os.cpu_count()))
    return out


def tiles_of(st):
    h, w, _ = st.shape
    hh, ww = h // 8 * 8, w // 8 * 8
    t = st[:hh, :ww].reshape(hh // 8, 8, ww // 8, 8, 17).transpose(0, 2, 1, 3, 4).reshape(-1, 64, 17)
    return t


def price_sync(t):
    rays = t[:, :, 1:]                       # [tile, lane, round]
    trips = rays.max(axis=1)                 # per round: the longest lane
    rounds = (rays > 0).any(axis=1).sum(axis=1)
    return trips.sum(axis=1) * C_STEP + rounds * C_SHADE


def price_gated(t, threshold):
    """Event simulation per tile (python loop over trips; only tiles with work)."""
    cost = np.zeros(len(t))
    for i, tile in enumerate(t):
        rays = tile[:, 1:]
        nr = (rays > 0).sum(axis=1)
        if nr.max() == 0:
            continue
        cur = np.zeros(64, np.int64)            # index of the ray a lane is on
        left = rays[:, 0].astype(np.int64)      # steps left in the current ray (0 = waiting for shading / done)
        walking = left > 0
        waiting = np.zeros(64, bool)
        c = C_SHADE  # ray generation
        while True:
            if walking.any():
                left[walking] -= 1
                c += C_STEP
                fin = walking & (left == 0)
                walking &= ~fin
                waiting |= fin
            if waiting.any() and (waiting.sum() >= threshold or not walking.any()):
                c += C_SHADE
                idx = np.nonzero(waiting)[0]
                cur[idx] += 1
                more = cur[idx] < nr[idx]
                nxt = idx[more]
                left[nxt] = rays[nxt, cur[nxt]]
                walking[nxt] = True
                waiting[idx] = False
            if not walking.any() and not waiting.any():
                break
        cost[i] = c
    return cost


def main():
    view = sys.argv[1] if len(sys.argv) > 1 else "bench"
    st = steps_for(view=view)
    t = tiles_of(st)
    rays = (t[:, :, 1:] > 0).sum()
    steps = t[:, :, 0].sum()
    print(f"view {view}: {len(t)} tiles, {rays} rays, {steps} steps ({steps / rays:.1f}/ray)")
    sync = price_sync(t)
    dense = steps / 64 * C_STEP + rays / 64 * C_SHADE
    print(f"sync : total {sync.sum() / 1e6:8.1f} M wave-instr, longest tile {sync.max() / 1e3:7.1f} k")
    print(f"dense: total {dense / 1e6:8.1f} M")
    # the gated simulation is slow in python: sample the tiles with work
    work = np.nonzero(t[:, :, 0].sum(axis=1) > 0)[0]
    rng = np.random.default_rng(1)
    pick = rng.choice(work, size=min(600, len(work)), replace=False)
    base = sync[pick].sum()
    for thr in (1, 8, 16, 24, 32, 48):
        g = price_gated(t[pick], thr)
        print(f"gated T={thr:2d}: {g.sum() / base:6.3f} x sync on {len(pick)} sampled tiles, longest {g.max() / 1e3:7.1f} k "
              f"(sync longest of the sample {sync[pick].max() / 1e3:7.1f} k)")


if __name__ == "__main__":
    main()


def decompose(view="bench"):
    st = steps_for(view=view)
    t = tiles_of(st)
    rays = t[:, :, 1:]
    trips = rays.max(axis=1)
    live = (rays > 0).sum(axis=1)
    rounds = (rays > 0).any(axis=1)
    print("round: tiles-in-round, wave trips (M), lane-steps (M), utilisation, mean live lanes")
    for r in range(rays.shape[2]):
        if not rounds[:, r].any():
            break
        tr = trips[:, r].sum()
        ls = rays[:, :, r].sum()
        print(f"  {r}: {rounds[:, r].sum():6d} {tr / 1e6:7.3f} {ls / 1e6:8.3f} {ls / (64 * tr):6.3f} {live[rounds[:, r], r].mean():6.1f}")
    print(f"walk {trips.sum() * C_STEP / 1e6:.1f} M, shade {rounds.sum() * C_SHADE / 1e6:.1f} M wave-instr")


def hybrid(view="bench", split_round=3):
    """rounds < split_round per 8x8 tile (monolithic), the paths still alive compacted (tile order) into dense waves of 64."""
    st = steps_for(view=view)
    t = tiles_of(st)
    rays = t[:, :, 1:]
    head = rays[:, :, :split_round]
    head_cost = head.max(axis=1).sum() * C_STEP + (head > 0).any(axis=1).sum() * C_SHADE
    alive = rays[:, :, split_round] > 0
    paths = rays[alive][:, split_round:]                      # [n_paths, rounds], tile order preserved
    n = len(paths)
    pad = (-n) % 64
    paths = np.concatenate([paths, np.zeros((pad, paths.shape[1]), paths.dtype)]).reshape(-1, 64, paths.shape[1])
    tail_cost = paths.max(axis=1).sum() * C_STEP + (paths > 0).any(axis=1).sum() * C_SHADE
    full = rays.max(axis=1).sum() * C_STEP + (rays > 0).any(axis=1).sum() * C_SHADE
    print(f"view {view} split at round {split_round}: {n} paths survive; monolithic {full / 1e6:.1f} M, hybrid {head_cost / 1e6:.1f} + {tail_cost / 1e6:.1f} "
          f"= {(head_cost + tail_cost) / 1e6:.1f} M wave-instr ({(head_cost + tail_cost) / full:.3f} x)")


def tail_refill(view="bench", split_round=3, ranges=(256, 1024), thresholds=(8, 16, 24, 32), c_small=90.0, sample_waves=300):
    """The compacted tail with per-lane refill: a wave owns `rng` consecutive queued paths; a lane whose path ended takes the
    next one; the advance block (shade a hit / start the bounce ray after a sun ray / finish) runs when >= T lanes wait for it
    or nobody walks.  Cost of an advance: C_SHADE if a lane in it shades, else c_small."""
    st = steps_for(view=view)
    t = tiles_of(st)
    rays = t[:, :, 1:]
    alive = rays[:, :, split_round] > 0
    paths = rays[alive][:, split_round:].astype(np.int64)     # [n, rounds]: sun, bounce, sun, bounce, ...
    n = len(paths)
    sync = paths_sync_cost(paths)
    print(f"{n} tail paths; lock-step chunks of 64: {sync / 1e6:.1f} M; dense bound: "
          f"{(paths.sum() / 64 * C_STEP + (paths > 0).sum() / 64 * (C_SHADE + c_small) / 2) / 1e6:.1f} M wave-instr")
    rs = np.random.default_rng(2)
    for rng in ranges:
        nw = n // rng
        pick = rs.choice(nw, size=min(sample_waves, nw), replace=False)
        for T in thresholds:
            total = 0.0
            for w in pick:
                total += refill_wave_cost(paths[w * rng:(w + 1) * rng], T, c_small)
            print(f"  range {rng:5d} T {T:2d}: {total / len(pick) * nw / 1e6:6.1f} M wave-instr")


def paths_sync_cost(paths):
    n = len(paths)
    pad = (-n) % 64
    p = np.concatenate([paths, np.zeros((pad, paths.shape[1]), paths.dtype)]).reshape(-1, 64, paths.shape[1])
    return p.max(axis=1).sum() * C_STEP + (p > 0).any(axis=1).sum() * C_SHADE


def refill_wave_cost(paths, T, c_small):
    npaths, nr = paths.shape
    nxt = 0
    path = -np.ones(64, np.int64)       # path index per lane
    ray = np.zeros(64, np.int64)        # index of the lane's current ray
    left = np.zeros(64, np.int64)
    state = np.zeros(64, np.int64)      # 0 idle, 1 walking, 2 waiting for the advance block
    cost = 0.0
    while True:
        walking = state == 1
        if walking.any():
            left[walking] -= 1
            cost += C_STEP
            fin = walking & (left <= 0)
            state[fin] = 2
        idle = state == 0
        want = (state == 2).sum() + (min(idle.sum(), npaths - nxt))
        if want > 0 and (want >= T or not (state == 1).any()):
            shades = False
            for l in range(64):
                if state[l] == 0 and nxt < npaths:
                    path[l] = nxt; nxt += 1; ray[l] = 0; shades = True          # new path: shade its hit, start the sun ray
                    left[l] = paths[path[l], 0]; state[l] = 1
                elif state[l] == 2:
                    ray[l] += 1
                    if ray[l] >= nr or paths[path[l], ray[l]] == 0:
                        state[l] = 0                                             # path finished (written out)
                        if nxt < npaths:
                            path[l] = nxt; nxt += 1; ray[l] = 0; shades = True
                            left[l] = paths[path[l], 0]; state[l] = 1
                    else:
                        if ray[l] % 2 == 0:
                            shades = True                                        # a bounce ray hit: shade, start the sun ray
                        left[l] = paths[path[l], ray[l]]; state[l] = 1
            cost += C_SHADE if shades else c_small
        if nxt >= npaths and not (state != 0).any():
            break
    return cost



def segment_pool(scene="menger", view="bench", bounces=4, w=1920, h=1080, split_round=3, c_step=116.0, c_io=80.0, ks=(1, 2, 3, 5, 8, 16), splits=((), (2,), (1, 2, 3, 4, 5, 6))):
    """Round 4: what would compacting the tail's live paths at EVERY path segment give — (a) inside a persistent wave that owns K chunks
    of 64 queued paths and re-packs its own survivors between segments (no launch boundary, no shared queue; c_io wave-instructions per
    pass for the record store / load), against (b) the shipped tail with a global compaction (a new launch) at the tail segments in
    `splits`.  Costs in wave-instructions from the oracle's per-ray step counts: c_step per lock-step trip (profiles/r04/walkf_step_isa.md:
    116 measured), C_SHADE per shading round."""
    st = steps_for(scene=scene, view=view, w=w, h=h, bounces=bounces)
    rays = tiles_of(st)[:, :, 1:]
    paths = rays[rays[:, :, split_round] > 0][:, split_round:].astype(np.int64)      # [n, rounds]: sun, bounce, sun, bounce, ...
    n, nr = paths.shape
    nseg = (nr + 1) // 2
    lane_steps = paths.sum()
    print(f"{scene} {view}, {bounces} bounces: {n} tail paths, {lane_steps / 1e6:.2f} M lane-steps")
    for K in ks:
        tot, trips = 0.0, 0
        for g0 in range(0, n, K * 64):
            p = paths[g0:g0 + K * 64]
            for j in range(nseg):
                s = p[:, 2 * j]
                b = p[:, 2 * j + 1] if 2 * j + 1 < nr else np.zeros(len(p), np.int64)
                a = s > 0 if j > 0 else np.ones(len(p), bool)
                if not a.any():
                    break
                sa, ba = s[a], b[a]
                for q in range(0, len(sa), 64):
                    ms, mb = sa[q:q + 64].max(), ba[q:q + 64].max()
                    tot += (ms + mb) * c_step + C_SHADE + (c_io if K > 1 else 0.0)
                    trips += ms + mb
        print(f"   a wave re-packs its own K = {K:2d} chunks at every segment: {tot / 1e6:6.1f} M wave-instr, lane utilisation {lane_steps / (trips * 64):.3f}")
    for split in splits:
        tot, trips, cur = 0.0, 0, paths
        for j in range(nseg):
            if j in split:
                cur = cur[cur[:, 2 * j] > 0]
                tot += len(cur) / 64 * 2 * c_io
            if len(cur) == 0:
                break
            pad = (-len(cur)) % 64
            P = np.concatenate([cur, np.zeros((pad, nr), np.int64)]).reshape(-1, 64, nr)
            s = P[:, :, 2 * j]
            b = P[:, :, 2 * j + 1] if 2 * j + 1 < nr else np.zeros_like(s)
            tot += (s.max(axis=1) + b.max(axis=1)).sum() * c_step + (s > 0).any(axis=1).sum() * C_SHADE
            trips += (s.max(axis=1) + b.max(axis=1)).sum()
        print(f"   shipped tail, global compaction (new launch) at tail segments {list(split)}: {tot / 1e6:6.1f} M, lane utilisation {lane_steps / (trips * 64):.3f}")


def steps_frames(frames, view="bench", w=1920, h=1080, bounces=4):
    """orc_trace_steps for `frames` consecutive frame numbers of a camera at rest: [frames, h, w, 17]."""
    pos, mrgb, size = scenes.load_scene("menger")
    cam = scenes.bench_camera(size) if view == "bench" else scenes.close_camera(size)
    octree = O.create_octree(pos, mrgb)
    noise = O.noise_table()
    u = O.Uniforms.default()
    u.set_camera(cam[0], O.camera_axis_scaled(cam[0], cam[1], cam[2], w, h))
    outs = []
    for f in range(frames):
        u.frame_number = 1 + f
        out = np.zeros((h, w, 17), np.int32)
        O.lib().orc_trace_steps(O._p(octree), O._p(noise), C.byref(u), C.c_int(bounces), C.c_int(0), C.c_int(0), C.c_int(w),
                                C.c_int(h), O._p(out), C.c_int(os.cpu_count()))
        outs.append(out)
    return np.stack(outs)


def lane_mappings(view="bench", frames=16, split=3):
    """Round 3: which 64 (pixel, frame) pairs share a wave of the head kernel.  A launch renders 16 frames of a camera at rest, so a
    wave could hold pw x ph pixels of pf frames instead of an 8 x 8 tile of one: the primary rays of a pixel are the same in every
    frame and its first sun rays nearly so.  Prices head (rounds < split, lock step per wave) + tail (compacted, chunks of 64)."""
    st = steps_frames(frames, view)[..., 1:]
    F, h, w, R = st.shape
    for pw, ph, pf in [(8, 8, 1), (8, 4, 2), (8, 2, 4), (8, 1, 8), (4, 2, 8), (4, 1, 16), (2, 2, 16), (4, 4, 4)]:
        hh, ww, ff = h // ph * ph, w // pw * pw, F // pf * pf
        a = st[:ff, :hh, :ww].reshape(ff // pf, pf, hh // ph, ph, ww // pw, pw, R).transpose(0, 2, 4, 1, 3, 5, 6).reshape(-1, 64, R)
        head = a[:, :, :split]
        hc = head.max(axis=1).sum() * C_STEP + (head > 0).any(axis=1).sum() * C_SHADE
        tc = paths_sync_cost(a[a[:, :, split] > 0][:, split:])
        util = [a[:, :, r].sum() / (64 * a[:, :, r].max(axis=1).sum()) for r in range(split)]
        print(f"{pw}x{ph} px x {pf} frames: head {hc / F / 1e6:6.1f} M (lane utilisation of its rounds {util[0]:.2f} {util[1]:.2f} {util[2]:.2f})"
              f"  tail {tc / F / 1e6:6.1f} M  total {(hc + tc) / F / 1e6:6.1f} M wave-instr per frame")


def branch_coherence(view="bench", split_round=3, frames=2, w=1920, h=1080, bounces=4):
    """Round 5: what would BRANCH-COHERENT tail chunks give?  A lock-step trip of the walk costs (the gfx950 listing,
    profiles/r04/walkf_step_isa.md) 32 VALU that every trip executes + 3 if any lane advances to a sibling + 57 if any lane descends + 53
    if any lane pops: 127 when the lanes of a wave split over all three, 35-90 when they agree.  The oracle logs the branch of every trip
    of every ray (orc_trace_branches); the tail's paths (alive at their second hit) are dealt to chunks of 64 in five ways and every
    chunk is priced trip by trip:
      shipped     shard = hash of the head wave's index (trace.hip: tail_shard), chunks = 64 consecutive records of a shard
      tile order  no hash: consecutive records in the head's wave order (a chunk = neighbouring tiles)
      cell        shard = the 4 x 4 x 4 cell of the hand-over hit inside the scene's box
      normal+cell shard = the face the path sits on (6) x the 2 x 2 x 2 cell: the sun rays of a chunk start on one face of one region
      oracle sort (bound) paths sorted by the branch string of their next ray — what no key known at the hand-over can reach
    Costs in VALU wave-instructions per frame, shading rounds (C_SHADE) included."""
    import ctypes as C
    pos, mrgb, size = scenes.load_scene("menger")
    cam = scenes.bench_camera(size) if view == "bench" else scenes.close_camera(size)
    octree = O.create_octree(pos, mrgb)
    noise = O.noise_table()
    u = O.Uniforms.default()
    u.set_camera(cam[0], O.camera_axis_scaled(cam[0], cam[1], cam[2], w, h))
    lo, hi = pos.min(0).astype(np.float64) * 0.5, (pos.max(0).astype(np.float64) + 1) * 0.5      # the scene's box in world units (voxel = 0.5)
    lo, hi = lo[[0, 2, 1]], hi[[0, 2, 1]]                                                            # voxel (x, y, z) -> world (x, z, y)
    B_ADV, B_DESC, B_POP = 3.0, 57.0, 53.0
    totals = {}
    for f in range(frames):
        u.frame_number = 1 + f
        st = np.zeros((h, w, 17), np.int32)
        O.lib().orc_trace_steps(O._p(octree), O._p(noise), C.byref(u), C.c_int(bounces), C.c_int(0), C.c_int(0), C.c_int(w), C.c_int(h), O._p(st),
                                C.c_int(os.cpu_count()))
        tot = st[..., 0].reshape(-1).astype(np.int64)
        offs = np.concatenate([[0], np.cumsum(tot)[:-1]]).astype(np.int64)
        flat = np.zeros(int(tot.sum()) + 16, np.uint8)
        O.lib().orc_trace_branches(O._p(octree), O._p(noise), C.byref(u), C.c_int(bounces), C.c_int(0), C.c_int(0), C.c_int(w), C.c_int(h), O._p(offs),
                                   O._p(flat), C.c_int(os.cpu_count()))
        rays = st[..., 1:].reshape(-1, 16).astype(np.int64)
        alive = np.nonzero(rays[:, split_round] > 0)[0]                       # pixels whose path reaches the tail
        nr = rays.shape[1] - split_round
        L = int(rays[alive][:, split_round:].max())
        n = len(alive)
        seq = np.full((n, nr, L), 255, np.uint8)                              # [path, tail round, trip] -> branch
        start = offs[alive] + rays[alive][:, :split_round].sum(1)
        for r in range(nr):
            ln = rays[alive, split_round + r]
            idx = start[:, None] + np.arange(L)[None, :]
            ok = np.arange(L)[None, :] < ln[:, None]
            seq[:, r, :][ok] = flat[idx[ok]] & 3            # (bits 2..: the node's level, see level_profile)
            start = start + ln
        # where and on which face each path is handed over: the origin of its first tail ray (hit + 1e-5 normal) from the oracle's ray log
        ys, xs = np.divmod(alive, w)
        org = np.zeros((n, 3)); nrm = np.zeros((n, 3))
        log = np.zeros((32, 12), np.float32)
        for i in range(n):
            k = O.lib().orc_trace_pixel_log(O._p(octree), O._p(noise), C.byref(u), C.c_int(bounces), C.c_int(int(xs[i])), C.c_int(int(ys[i])), O._p(log))
            assert k > split_round
            org[i] = log[split_round, 0:3]; nrm[i] = log[split_round - 1, 9:12]      # the ray before it (bounce ray of segment 1) hit with this normal
        cell4 = np.clip(((org - lo) / (hi - lo) * 4).astype(int), 0, 3)
        cell2 = cell4 // 2
        face = np.argmax(np.abs(nrm), 1) * 2 + (nrm[np.arange(n), np.argmax(np.abs(nrm), 1)] > 0)
        # the head's wave of every path: 8 x 8 pixel tiles in raster order (the cost order permutes waves, not a wave's contents)
        wave = (ys // 8) * (w // 8) + xs // 8

        def price(order):
            s = seq[order]
            pad = (-len(s)) % 64
            s = np.concatenate([s, np.full((pad, nr, L), 255, np.uint8)]).reshape(-1, 64, nr, L)
            live = (s != 255).any(1)                                         # [chunk, round, trip]: somebody is on this trip
            cost = (32.0 * live + B_ADV * (s == 0).any(1) + B_DESC * (s == 1).any(1) + B_POP * (s == 2).any(1)).sum()
            trips = live.sum()
            return cost + live.any(2).sum() * C_SHADE, trips, cost / max(trips, 1)

        def by_shard(key, nshards):
            # records of a shard in the head's wave order; chunks never mix shards (the last chunk of a shard is part-filled)
            out = []
            for sh in range(nshards):
                m = np.nonzero(key == sh)[0]
                m = m[np.argsort(wave[m], kind="stable")]
                out.append(m)
                pad = (-len(m)) % 64
                out.append(np.full(pad, -1))
            return np.concatenate(out)

        deals = {"shipped (hash of the head wave)": by_shard(((wave.astype(np.uint64) * 0x9E3779B1) & 0xffffffff) >> 26, 64),
                 "tile order (no hash)": np.argsort(wave, kind="stable"),
                 "4x4x4 cell of the hand-over hit": by_shard(cell4[:, 0] * 16 + cell4[:, 1] * 4 + cell4[:, 2], 64),
                 "face x 2x2x2 cell": by_shard(face * 8 + cell2[:, 0] * 4 + cell2[:, 1] * 2 + cell2[:, 2], 48),
                 "oracle sort by the next ray's branch string (bound)": np.lexsort(seq[:, 0, ::-1].T)}
        seq_pad = np.concatenate([seq, np.full((1, nr, L), 255, np.uint8)])       # index -1 -> an idle lane
        for name, order in deals.items():
            s_order = np.where(order < 0, n, order)
            c, trips, per = price_padded(seq_pad, s_order, nr, L, B_ADV, B_DESC, B_POP)
            t = totals.setdefault(name, [0.0, 0, 0.0, 0])
            t[0] += c; t[1] += trips; t[3] += (order >= 0).sum()
        print(f"frame {f + 1}: {n} tail paths, {int((seq != 255).sum())} lane-trips", flush=True)
    base = totals["shipped (hash of the head wave)"][0]
    for name, (c, trips, _, lanes) in totals.items():
        walk = c - 0
        print(f"  {name:55s}: {c / frames / 1e6:7.2f} M VALU wave-instr per frame ({c / base:5.3f} x), {trips / frames / 1e3:7.1f} k wave-trips")
    return totals


def price_padded(seq_pad, order, nr, L, b_adv, b_desc, b_pop):
    s = seq_pad[order]
    pad = (-len(s)) % 64
    if pad:
        s = np.concatenate([s, np.full((pad, nr, L), 255, np.uint8)])
    s = s.reshape(-1, 64, nr, L)
    live = (s != 255).any(1)
    cost = (32.0 * live + b_adv * (s == 0).any(1) + b_desc * (s == 1).any(1) + b_pop * (s == 2).any(1)).sum()
    trips = int(live.sum())
    return cost + live.any(2).sum() * C_SHADE, trips, cost / max(trips, 1)


def trip_sort(view="bench", split_round=3):
    """Round 6: what is trip-count COHERENCE worth in the compacted tail?  The tail's lanes run at 28 % (31 % of the trips of a wave's longest
    lane, on the oracle's step counts): a wave waits for its slowest cast, cast after cast.  If the 64 paths of a chunk had similar trip counts
    — an ORACLE sort: no kernel knows a cast's length beforehand — the tail would cost:
        tile order (ships) 36.5 M wave-instructions; sorted by a path's total trips 0.72 x; by its first cast 0.92 x; by its first, then its
        second cast 0.75 x; in random order 1.12 x; re-sorted before EVERY cast (steps only) 0.45 x = the dense bound.
    So the prize for a predictor of cast length is 12 % of the stage (total-trips sort) to 25 % (per cast) — and round 2's priced answer to
    "run a cast for K trips, hand the stragglers over" (0.64-0.73 x the trips) is the buildable form of it, at 160 bytes of walk state per
    straggler.  usage: python -c "import sim_schedule as S; S.trip_sort()" """
    st = steps_for(view=view)
    rays = tiles_of(st)[:, :, 1:]
    alive = rays[:, :, split_round] > 0
    paths = rays[alive][:, split_round:]
    base = paths_sync_cost(paths)
    rng = np.random.default_rng(1)
    for label, order in (("tile order (ships)", np.arange(len(paths))), ("oracle: by a path's total trips", np.argsort(-paths.sum(1), kind="stable")),
                         ("oracle: by the first cast's trips", np.argsort(-paths[:, 0], kind="stable")),
                         ("oracle: by the first, then the second cast", np.lexsort((-paths[:, 1], -paths[:, 0]))), ("random order", rng.permutation(len(paths)))):
        c = paths_sync_cost(paths[order])
        print(f"{label:46s} {c / 1e6:8.2f} M wave-instructions ({c / base:.3f} x)")
    pad = (-len(paths)) % 64
    pp = np.concatenate([paths, np.zeros((pad, paths.shape[1]), paths.dtype)]).reshape(-1, 64, paths.shape[1])
    tot = 0
    for r in range(paths.shape[1]):
        col = np.sort(paths[:, r][paths[:, r] > 0])[::-1]
        col = np.concatenate([col, np.zeros((-len(col)) % 64, col.dtype)]).reshape(-1, 64)
        tot += col.max(axis=1).sum()
    print(f"lane utilisation in trips, tile order: {pp.sum() / (pp.max(axis=1).sum() * 64):.3f}; steps only: tile order {pp.max(axis=1).sum() * C_STEP / 1e6:.1f} M, "
          f"re-sorted before every cast {tot * C_STEP / 1e6:.1f} M, dense {paths.sum() / 64 * C_STEP / 1e6:.1f} M")


def level_profile(view="bench", w=1920, h=1080, bounces=4):
    """Round 5: where in the tree do the walk's trips happen?  Per node level (0 = the root) the advance / descend / pop trips of every ray of
    the bench frame, and what share of the walk's lane-level VALU cost (32 + 3 / 57 / 53 per trip) falls into the bottom two node levels —
    the part a 4^3 bit-mask brick with a unit-step DDA would replace (HISTORY section 11: the traversal structure is free, only plane times
    and the descend's position test are contract)."""
    import ctypes as C
    pos, mrgb, size = scenes.load_scene("menger")
    cam = scenes.bench_camera(size) if view == "bench" else scenes.close_camera(size)
    octree = O.create_octree(pos, mrgb)
    depth = O.voxel_depth(pos)
    noise = O.noise_table()
    u = O.Uniforms.default()
    u.set_camera(cam[0], O.camera_axis_scaled(cam[0], cam[1], cam[2], w, h))
    u.frame_number = 1
    st = np.zeros((h, w, 17), np.int32)
    O.lib().orc_trace_steps(O._p(octree), O._p(noise), C.byref(u), C.c_int(bounces), C.c_int(0), C.c_int(0), C.c_int(w), C.c_int(h), O._p(st), C.c_int(os.cpu_count()))
    tot = st[..., 0].reshape(-1).astype(np.int64)
    offs = np.concatenate([[0], np.cumsum(tot)[:-1]]).astype(np.int64)
    flat = np.zeros(int(tot.sum()) + 16, np.uint8)
    O.lib().orc_trace_branches(O._p(octree), O._p(noise), C.byref(u), C.c_int(bounces), C.c_int(0), C.c_int(0), C.c_int(w), C.c_int(h), O._p(offs), O._p(flat),
                               C.c_int(os.cpu_count()))
    flat = flat[:int(tot.sum())]
    kind, level = flat & 3, flat >> 2
    cost = np.array([35.0, 89.0, 85.0, 32.0])
    total = cost[kind].sum()
    print(f"view {view}: {len(flat)} lane-trips, tree depth {depth} (node levels 0..{depth}); lane-level cost {total / 1e6:.1f} M VALU")
    print("level: advance  descend  pop   (share of the cost)")
    for l in range(depth + 1):
        m = (level == l) & (kind < 3)
        if not m.any():
            continue
        a, d, p = (int(((kind == k) & m).sum()) for k in (0, 1, 2))
        print(f"  {l}: {a:9d} {d:9d} {p:9d}   {cost[kind[m]].sum() / total * 100:5.1f} %")
    bottom = (level >= depth - 1) & (kind < 3)
    print(f"bottom two node levels ({depth - 1}, {depth}): {cost[kind[bottom]].sum() / total * 100:.1f} % of the cost, {bottom.mean() * 100:.1f} % of the trips; "
          f"of it descends + pops {cost[kind[bottom & (kind > 0)]].sum() / total * 100:.1f} %")
    # what a unit-step DDA inside a 4^3 mask would pay for the same stretch: one step per advance trip there (25 VALU), nothing for the
    # descends and pops inside, one entry per descend INTO level depth-1
    enters = ((level == depth - 2) & (kind == 1)).sum()
    dda = ((kind == 0) & bottom).sum() * 25.0 + enters * 60.0
    print(f"a 4^3-mask DDA for that stretch: {dda / total * 100:.1f} % of today's cost  ->  walk cost x {(total - cost[kind[bottom]].sum() + dda) / total:.3f}")
