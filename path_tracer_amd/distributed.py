"""Multi-GPU rendering: one process per GPU, torch.distributed (backend "nccl" = RCCL over xGMI).

Replaces the reference's only work split — contiguous pixel chunks over OS threads
(multithreading.zig:40-44, main.zig:142-165) — with the two splits of a node of GPUs:

  "samples"  every rank traces samples [g*S/G, (g+1)*S/G) of EVERY pixel into a full-frame
             fp32 sum buffer; one reduce(SUM) to rank 0 over RCCL adds the G buffers.  For
             high-spp frames (BASELINE config 4).  Philox streams are keyed by the global
             sample index, so the image does not depend on G beyond fp32 add order.
  "tiles"    the frame is cut into runs of `tile_pixels` consecutive pixels, dealt round-robin
             to the ranks (sky rows are cheap: contiguous bands would be unbalanced); each
             rank leaves the other ranks' pixels at 0, so the same reduce(SUM) assembles the
             frame exactly (x + 0 = x): identical bits to a single-GPU render.  For large
             frames / scenes (BASELINE config 5).

The path has exactly one exchange step — that reduce; nothing else is communicated (the scene
is a few KB..MB and is replicated).
"""
import torch
import torch.distributed as dist

from .host import make_pt_config


def sample_range(total_samples, world, rank):
    """[g*S/G, (g+1)*S/G) — SURVEY.md §8(e)."""
    return (rank * total_samples) // world, ((rank + 1) * total_samples) // world


def plan(img, rt, seed, world, rank, split="samples", chunk=0, deterministic=False, tile_pixels=256):
    """The pt_config rank `rank` of `world` renders, or None when its share is empty (more ranks than samples/tiles).
    The split itself is the library's (ptgpu_multi_plan, include/ptgpu.h): the one-process ptgpu_multi_* path and this
    one-process-per-GPU path partition a frame identically."""
    from .gpu import multi_plan
    whole = make_pt_config(img, rt, seed, deterministic=deterministic, chunk=chunk,
                           tile_pixels=tile_pixels if split == "tiles" else 0)
    if split not in ("samples", "tiles"):
        raise ValueError("split must be 'samples' or 'tiles'")
    return multi_plan(whole, split, world, rank)


def render_sum_distributed(render_into, img, rt, seed, split="samples", chunk=0, deterministic=False, tile_pixels=256,
                           device="cuda", group=None, frame=None, everywhere=False):
    """Every rank renders its share into a zeroed full-frame buffer and the buffers are summed
    on rank 0 (dist.reduce, RCCL for CUDA tensors).  `render_into(cfg, frame_tensor)` is the
    single-GPU render (GpuContext.render_sum_device on frame.data_ptr()).  Returns the frame
    tensor (complete on rank 0 only; on every rank with everywhere=True, which uses all_reduce)."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    if frame is None:
        frame = torch.zeros(img.height * img.width * 3, dtype=torch.float32, device=device)
    else:
        frame.zero_()
    cfg = plan(img, rt, seed, world, rank, split, chunk, deterministic, tile_pixels)
    if cfg is not None:
        render_into(cfg, frame)
    if world > 1:
        if everywhere:
            dist.all_reduce(frame, op=dist.ReduceOp.SUM, group=group)
        else:
            dist.reduce(frame, dst=0, op=dist.ReduceOp.SUM, group=group)
    return frame


def gpu_render_into(gpu, sampler=0):
    """render_into for a GpuContext: enqueue on torch's current stream, device memory only.
    sampler: pt_config.sampler of every render (PT_SAMPLER_REJECTION, the reference's loops, unless asked otherwise)."""

    def fn(cfg, frame):
        assert frame.is_cuda and frame.dtype == torch.float32 and frame.is_contiguous()
        cfg.sampler = int(sampler)
        gpu.render_sum_device(cfg, frame.data_ptr(), torch.cuda.current_stream().cuda_stream)

    return fn
