"""Headless frame loop: what src/main.rs + Context::update/render do interactively, without
the window — a camera path, N frames of trace -> temporal -> denoise, and image dumps.

    python -m gpu_voxel_raytracer_amd.frame_loop --scene menger --frames 32 --radius 2 --out gpurun_out/menger

Scenes: a fixture name (tests/golden/scenes/*.npz), a .vox file (`--whole-scene`: every model of its scene graph),
`menger:<level>[:clip[:emissive_period]]`, or `default[:seed]` = the reference's start-up scene and camera
(src/context.rs:838-910, 618-622).  `--noise blue` generates the blue-noise table on the GPU, `--noise <file.zip>` loads
one in the reference's resource format.
The reference shows `denoised_color` through an sRGB swap chain without tone mapping (shaders/display.frag,
src/context.rs:1352-1403); the PNGs are written the same way (clamp to [0,1], sRGB encode)."""
import argparse
import os

import numpy as np

from . import ALL, DENOISED, Camera, Context, scenes


def orbit_camera(size_xyz, t, radius_scale=1.5, height=0.6, fov=scenes.FOV_70):
    """Camera on a circle around the model centre, t in [0,1) = one revolution (a stand-in for the reference's
    WASD fly camera, src/context.rs:1959-2001)."""
    ext = scenes.world_extent(size_xyz)
    c = ext * np.float32(0.5)
    e = np.float32(ext.max())
    a = np.float32(2 * np.pi * t)
    position = (c + e * np.array([radius_scale * np.cos(a), height, radius_scale * np.sin(a)], np.float32)).astype(np.float32)
    return position, (c - position).astype(np.float32), fov


def srgb8(rgb):
    """Linear -> 8-bit sRGB, as a Bgra8UnormSrgb swap chain stores it (src/context.rs:696-706)."""
    x = np.clip(np.nan_to_num(rgb, nan=0.0, posinf=1.0, neginf=0.0), 0.0, 1.0)
    y = np.where(x <= 0.0031308, 12.92 * x, 1.055 * np.power(x, 1 / 2.4) - 0.055)
    return (y * 255.0 + 0.5).astype(np.uint8)


def load_into(ctx, scene, whole_scene=False):
    """Returns the model size (x, y, z in file axes) used for camera placement (None: keep the reference's start camera)."""
    from . import host
    if scene.startswith("default"):
        seed = int(scene.split(":")[1]) if ":" in scene else 1
        ctx.recreate_octree(*host.default_scene_voxels(seed))
        return None
    if scene.startswith("menger:"):
        parts = [int(p) for p in scene.split(":")[1:]] + [0, 0]
        level, clip, period = parts[0], parts[1], parts[2]
        ctx.set_menger(level, clip, (0, 0x7b, 0xa2, 0x3f), period)
        side = min(3 ** level, clip or 3 ** level)
        return (side, side, side)
    if os.path.exists(scene):
        data = open(scene, "rb").read()
        if whole_scene:
            pos, mrgb, (lo, hi) = host.vox_scene_to_voxels(data, host.VOX_ALL_MODELS | host.VOX_LENIENT_MATERIALS | host.VOX_REBASE)
            size = (hi[0] + 1, hi[2] + 1, hi[1] + 1)   # renderer axes (x, z, y) -> file axes
        else:
            pos, mrgb, size = host.vox_to_voxels(data)
        ctx.recreate_octree(pos, mrgb)
        return size
    pos, mrgb, size = scenes.load_scene(scene)
    ctx.recreate_octree(pos, mrgb)
    return size


def run(scene="menger", width=1280, height=720, frames=16, bounces=3, radius=0, moving=False, out=None, device=0,
        frames_in_flight=1, dump_every=0, noise="white", spp=1, whole_scene=False, float_dump=False):
    """Renders `frames` frames; returns the last denoised frame (float32 [h, w, 4]) and the context statistics.
    float_dump: also write the frame losslessly as <out>.exr (OpenEXR, 32-bit float, save_exr) and <out>.npy (float32 [h, w, 4]):
    linear radiance as denoise.comp stores it — the lossless counterparts of the 8-bit sRGB PNG (PNG / EXR)."""
    from . import host
    with Context(width, height, device=device, max_bounces=bounces, frames_in_flight=frames_in_flight,
                 frames_per_launch=min(max(spp, 1), 32) if spp > 1 else min(max(frames, 1), 16)) as ctx:
        if noise == "blue":
            ctx.set_noise(host.blue_noise(device=device))
        elif noise != "white":
            size_px, table = host.load_blue_noise(noise)
            if size_px != 128:
                raise ValueError("blue noise images must be 128 x 128 (src/context.rs:1027-1032)")
            ctx.set_noise(table)
        size = load_into(ctx, scene, whole_scene)
        ctx.denoise_uniforms.radius = radius
        if spp == 1 and size is not None and not (out and dump_every):
            # nothing to dump in between: the whole camera path in one call, several frames per trace launch
            path = [orbit_camera(size, 0.62 + ((f / max(frames, 1)) * 0.25 if moving else 0.0)) for f in range(frames)]
            ctx.render_path(ALL, [p[0] for p in path], [p[1] for p in path], path[0][2])
            frames = 0
        for f in range(frames):
            t = (f / max(frames, 1)) * 0.25 if moving else 0.0
            ctx.camera = Camera(*orbit_camera(size, 0.62 + t)) if size is not None else Camera()
            if spp > 1:
                ctx.render_spp(ALL, spp)
            else:
                ctx.render(ALL)
            if out and dump_every and (f + 1) % dump_every == 0:
                save_png(ctx.read(DENOISED), f"{out}_{f + 1:04d}.png")
        img = ctx.read(DENOISED)
        st = ctx.stats()
    if out:
        save_png(img, out + ".png")
        if float_dump:
            np.save(out + ".npy", np.ascontiguousarray(img, np.float32))
            save_exr(img, out + ".exr")
    return img, st


def save_exr(img, path):
    """The frame as an OpenEXR file (SURVEY.md 8f n1 asks for PNG / EXR): linear radiance, 32-bit float channels A, B, G, R, one
    uncompressed scan line per block, written from the file-layout document of OpenEXR 2 with numpy (no EXR library is in this image).
    img: float32 [h, w, 4] = (r, g, b, a) as denoise.comp stores it."""
    import struct
    img = np.ascontiguousarray(img, np.float32)
    h, w = img.shape[:2]

    def attr(name, kind, value):
        return name.encode() + b"\0" + kind.encode() + b"\0" + struct.pack("<i", len(value)) + value
    chlist = b"".join(c + b"\0" + struct.pack("<iB3xii", 2, 0, 1, 1) for c in (b"A", b"B", b"G", b"R")) + b"\0"   # pixel type 2 = FLOAT
    window = struct.pack("<4i", 0, 0, w - 1, h - 1)
    header = (struct.pack("<ii", 20000630, 2) + attr("channels", "chlist", chlist) + attr("compression", "compression", b"\0") +
              attr("dataWindow", "box2i", window) + attr("displayWindow", "box2i", window) + attr("lineOrder", "lineOrder", b"\0") +
              attr("pixelAspectRatio", "float", struct.pack("<f", 1.0)) + attr("screenWindowCenter", "v2f", struct.pack("<2f", 0.0, 0.0)) +
              attr("screenWindowWidth", "float", struct.pack("<f", 1.0)) + b"\0")
    row_bytes = 4 * w * 4
    first = len(header) + 8 * h
    offsets = (first + np.arange(h, dtype=np.uint64) * np.uint64(8 + row_bytes)).astype("<u8")
    planes = np.ascontiguousarray(img[:, :, [3, 2, 1, 0]].transpose(0, 2, 1)).astype("<f4")      # [row][A, B, G, R][x]
    os.makedirs(os.path.dirname(os.path.abspath(path)) or ".", exist_ok=True)
    with open(path, "wb") as f:
        f.write(header)
        f.write(offsets.tobytes())
        for y in range(h):
            f.write(struct.pack("<ii", y, row_bytes))
            f.write(planes[y].tobytes())


def save_png(img, path):
    from PIL import Image
    os.makedirs(os.path.dirname(os.path.abspath(path)) or ".", exist_ok=True)
    Image.fromarray(srgb8(img[..., :3])).save(path)


def main():
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("--scene", default="menger")
    ap.add_argument("--width", type=int, default=1280)
    ap.add_argument("--height", type=int, default=720)
    ap.add_argument("--frames", type=int, default=16)
    ap.add_argument("--bounces", type=int, default=3)
    ap.add_argument("--radius", type=int, default=0)
    ap.add_argument("--moving", action="store_true", help="orbit the camera (temporal reprojection at work)")
    ap.add_argument("--dump-every", type=int, default=0)
    ap.add_argument("--noise", default="white", help="white (seeded stand-in), blue (void-and-cluster, made on the GPU) or an archive")
    ap.add_argument("--spp", type=int, default=1, help="samples per pixel per displayed frame (vxrt_render_spp)")
    ap.add_argument("--whole-scene", action="store_true", help="place every model of a .vox file's scene graph")
    ap.add_argument("--float-dump", action="store_true", help="also write <out>.exr (OpenEXR, float32) and <out>.npy: the frame as linear radiance, lossless")
    ap.add_argument("--out", default="gpurun_out/frame")
    args = ap.parse_args()
    img, st = run(args.scene, args.width, args.height, args.frames, args.bounces, args.radius, args.moving, args.out,
                  dump_every=args.dump_every, noise=args.noise, spp=args.spp, whole_scene=args.whole_scene, float_dump=args.float_dump)
    print(f"{args.scene}: {st.frames} frames, {st.rays} rays, image {img.shape[1]}x{img.shape[0]} -> {args.out}.png, "
          f"mean radiance {float(np.nanmean(img[..., :3])):.4f}")


if __name__ == "__main__":
    main()
