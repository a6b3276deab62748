// trace_tail_body.h — what bounce_kernel does with ONE queued path record (one lane): path segments first_bounce .. last_bounce in lock step
// with the wave's other lanes.  Shared by bounce_kernel (trace_tail.hip) and the fused kernel (trace.hip: fused_kernel), so that both
// run the same operations in the same order.
#pragma once
#include "trace_common.h"

namespace vxrt {

// rec: the path as it was handed over (ambient_rays in bits 8-15 of normal_ambient; the bits above belong to the queue);
// shard: where a path that survives last_bounce is appended in `out`.
template <bool kWide>
__device__ __forceinline__ void bounce_path(const TraceArgs& a, const Caster<kWide>& caster, PathRec rec, const PathQueue& out, unsigned shard, int first_bounce,
                                            int last_bounce, f3 sun_dir, f3 sun_color, f3 sky, uint32_t& rays) {
        Rng rng;
        rng.noise = a.noise;
        rng.index = rec.rng_index;
        // Path segments first_bounce .. last_bounce run in this launch (lanes whose path ends simply idle);
        // a path that is still alive after segment last_bounce goes to the next queue.
        for (int bounce = first_bounce;; bounce++) {
            const f3 n = mk3(unpack_axis(rec.normal_ambient & 3u), unpack_axis((rec.normal_ambient >> 2) & 3u), unpack_axis((rec.normal_ambient >> 4) & 3u));
            uint32_t ambient_rays = (rec.normal_ambient >> 8) & 0xffu;
            f3 sample = rec.sample, blend = rec.blend;
            const f3 color = bounce == 0 ? splat3(1.0f) : node_color(rec.node);          // voxels.comp:317
            const f3 emit = node_emittance(rec.node, a.emit_strength);
            const f3 o = rec.hit_pos + 1e-5f * n;                                       // voxels.comp:333,353,370
            f3 d;
            if (rng.next() < a.specularity) {  // specular                              voxels.comp:326-334
                d = norm3(reflect3(rec.dir, n));
                sample = sample + emit * blend;
                blend = blend * ((2.0f * color) * dot3(d, n));
            } else if (a.sun_strength > 0.0f) {  // diffuse + sun sample                  voxels.comp:339-371
                float r0 = rng.next(), r1 = rng.next(), r2 = rng.next();
                f3 up_dir = norm3(cross3(mk3(r0, r1, r2), sun_dir));
                f3 right_dir = norm3(cross3(sun_dir, up_dir));
                float dx = 2.0f * rng.next() - 1.0f;
                float dy = 2.0f * rng.next() - 1.0f;
                f3 light_dir = ld3(a.sun_dir_n) + (dx * right_dir + dy * up_dir) * a.sun_size;
                f3 to_light = norm3(-light_dir);
                ambient_rays++;
                const float facing = vx_max(0.0f, dot3(n, to_light));
                RayHit sun_hit;
                rays++;
                const bool lit = !caster.cast(o, to_light, sun_hit);
                // normal, colour and emittance are derived again from the record instead of being kept alive across the cast
                // (the same operations on the same operands; ten registers fewer while the sun ray walks)
#ifndef VXRT_TAIL_REMAT
#define VXRT_TAIL_REMAT 1
#endif
#if VXRT_TAIL_REMAT
                uint32_t packed = rec.normal_ambient;
                int32_t word = rec.node;
                asm volatile("" : "+v"(packed), "+v"(word));   // new values to the compiler: no common subexpression with n / color / emit above
                const f3 n2 = mk3(unpack_axis(packed & 3u), unpack_axis((packed >> 2) & 3u), unpack_axis((packed >> 4) & 3u));
                const f3 color2 = bounce == 0 ? splat3(1.0f) : node_color(word);
                const f3 emit2 = node_emittance(word, a.emit_strength);
#else
                const f3 n2 = n, color2 = color, emit2 = emit;
#endif
                if (lit) sample = sample + ((sun_color * color2) * blend) * facing;
                d = random_hemisphere(n2, rng);
                sample = sample + emit2 * blend;
                blend = blend * (color2 * dot3(n2, d));
            } else {  // diffuse, sun switched off
                d = random_hemisphere(n, rng);
                sample = sample + emit * blend;
                blend = blend * (color * dot3(n, d));
            }

            bool finished = true;
            if (bounce + 1 < a.max_bounces) {  // next path segment                        voxels.comp:309-313
                RayHit hit;
                rays++;
                if (caster.cast(o, d, hit)) {
                    const f3 hn = hit.normal;
                    rec.hit_pos = o + d * hit.time;
                    rec.node = hit.node;
                    rec.dir = d;
                    rec.normal_ambient = pack_axis(hn.x) | pack_axis(hn.y) << 2 | pack_axis(hn.z) << 4 | ambient_rays << 8;
                    rec.sample = sample;
                    rec.blend = blend;
                    finished = false;
                } else {
                    sample = sample + sky * blend;                                        // voxels.comp:384
                }
            }
            if (finished) {
                f3 outc = sample / float(ambient_rays);                                   // voxels.comp:391
                store_out(a.out[rec.pix >> kPixBits].color + (rec.pix & ((1u << kPixBits) - 1u)), make_float4(outc.x, outc.y, outc.z, 1.0f));
                break;
            }
            // compact again: to the next queue — unless that is full, then this lane simply goes on
            if (bounce == last_bounce) {
                const uint32_t slot = queue_reserve(out, shard);
                if (slot != kNoSlot) {
                    rec.rng_index = rng.index;
                    queue_store(out, shard, slot, rec);
                    break;
                }
            }
        }
}

}  // namespace vxrt
