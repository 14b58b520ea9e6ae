"""Image-shard data parallelism: images are independent, models are replicated, rank r
owns images r, r+world, ...; the only exchange is ONE gather of fixed-stride detection records.
One process per GPU, torch.distributed ("nccl" == RCCL on ROCm; "gloo" in the CPU tests)."""
import numpy as np
import torch
import torch.distributed as dist

RECORD_FIELDS = 8  # image_id, detector_id, cx, cy, w, h, score, probability


def shard_indices(n_items, rank, world):
    """image i -> rank (i mod world)"""
    return list(range(rank, n_items, world))


def pack_records(image_ids, detector_ids, dets):
    """dets: structured array with cx, cy, w, h, score, probability -> float64 [n, 8]"""
    n = len(dets)
    out = np.zeros((n, RECORD_FIELDS), np.float64)
    if n:
        out[:, 0] = image_ids
        out[:, 1] = detector_ids
        for j, f in enumerate(("cx", "cy", "w", "h", "score", "probability")):
            out[:, 2 + j] = dets[f]
    return out


def gather_records(local, cap, device="cpu", group=None):
    """All ranks contribute up to `cap` records; every rank receives all of them, ordered by
    (image_id, detector_id, original order).  One all_gather of a padded [cap+1, 8] buffer (row 0 holds
    the count) -- payload is KBs, latency bound, so callers batch many images per gather."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    n = min(len(local), cap)
    buf = torch.zeros((cap + 1, RECORD_FIELDS), dtype=torch.float64, device=device)
    buf[0, 0] = float(len(local))
    if n:
        buf[1:n + 1] = torch.from_numpy(np.ascontiguousarray(local[:n])).to(device)
    if world == 1:
        parts = [buf]
    else:
        parts = [torch.empty_like(buf) for _ in range(world)]
        dist.all_gather(parts, buf, group=group)
    rows, truncated = [], False
    for p in parts:
        p = p.cpu().numpy()
        cnt = int(p[0, 0])
        truncated |= cnt > cap
        rows.append(p[1:min(cnt, cap) + 1])
    allr = np.concatenate(rows) if rows else np.zeros((0, RECORD_FIELDS))
    order = np.lexsort((np.arange(len(allr)), allr[:, 1], allr[:, 0]))
    return allr[order], truncated
