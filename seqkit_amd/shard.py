"""Multi-GPU decomposition of the hot path (one process per GPU).

Records are independent, so a batch is cut into contiguous record ranges of about equal BYTE
volume, one per rank; reverse-complement and per-record counts then need no exchange at all
(each rank's outputs are a contiguous slice of the result).  A record longer than a rank's
share is cut INSIDE the record: input slice [a, b) of a record of length L is itself a record
whose reverse-complement lands at [L-b, L-a) of the result, so slices are independent too.
The only collective of the path is the element-wise reduction of the five `stats` totals
(seqkit/cmd/stat.go:106-125): sum for num / len_sum / gap_sum, min / max for the extremes.
"""
import numpy as np


def split_by_bytes(offsets, nparts):
    """Contiguous record ranges [(lo, hi)) * nparts with byte volumes as equal as record
    boundaries allow (greedy prefix split of the CSR offsets)."""
    offsets = np.asarray(offsets, dtype=np.uint64)
    nrec = len(offsets) - 1
    total = int(offsets[-1])
    cuts = [0]
    for k in range(1, nparts):
        target = total * k // nparts
        r = int(np.searchsorted(offsets, np.uint64(target), side="left"))
        # nearer of the two record boundaries around the byte target
        if r > 0 and r <= nrec and abs(int(offsets[r - 1]) - target) <= abs(int(offsets[min(r, nrec)]) - target):
            r -= 1
        cuts.append(min(max(r, cuts[-1]), nrec))
    cuts.append(nrec)
    return [(cuts[i], cuts[i + 1]) for i in range(nparts)]


def split_record(length, nparts, align=16):
    """Cuts one long record into `nparts` slices at `align`-byte multiples.
    Returns [(a, b, out_pos)]: input slice [a, b) and where its reverse goes in the output."""
    cuts = [0]
    for k in range(1, nparts):
        c = (length * k // nparts) // align * align
        cuts.append(max(c, cuts[-1]))
    cuts.append(length)
    return [(cuts[i], cuts[i + 1], length - cuts[i + 1]) for i in range(nparts)]


def split_plane(offsets, nparts):
    """Byte positions cuts[0..nparts] that cut the plane of a batch into shares of about equal volume
    (the rule of the library's skh_split_plane): a cut lies at the record boundary nearest to k/nparts
    of the bytes when that boundary is within 1/16 of a share of it, else inside the record at a multiple
    of 16 bytes from the record's start."""
    offsets = np.asarray(offsets, dtype=np.uint64)
    nrec = len(offsets) - 1
    total = int(offsets[-1])
    tol = total // nparts // 16
    cuts = [0]
    for k in range(1, nparts):
        target = total * k // nparts
        r = int(np.searchsorted(offsets, np.uint64(target), side="left"))      # offsets[r] >= target
        cut = int(offsets[min(r, nrec)])
        if cut != target:
            s, e = int(offsets[r - 1]), int(offsets[r])
            if e - target <= tol and e - target <= target - s:
                cut = e
            elif target - s <= tol:
                cut = s
            else:
                cut = s + (target - s) // 16 * 16
        cuts.append(max(cut, cuts[-1]))
    cuts.append(total)
    return cuts


def pieces_of_share(offsets, b0, b1):
    """The pieces [(record, a, b)] that tile plane bytes [b0, b1): record-relative slices, whole records
    as (r, 0, L).  Empty records are not listed."""
    offsets = np.asarray(offsets, dtype=np.uint64)
    out = []
    if b1 <= b0:
        return out
    r = int(np.searchsorted(offsets, np.uint64(b0), side="right")) - 1
    while r < len(offsets) - 1 and int(offsets[r]) < b1:
        s, e = int(offsets[r]), int(offsets[r + 1])
        a, b = max(s, b0), min(e, b1)
        if b > a:
            out.append((r, a - s, b - s))
        r += 1
    return out


def plan_pieces(lengths, nparts, align=16):
    """Byte-volume sharding with cuts inside records: rank k gets the pieces [(record, a, b)] that
    cover bytes [k * total / nparts, (k+1) * total / nparts) of the batch, a cut inside a record moved
    down to a multiple of `align` within that record.  A piece (r, a, b) is a record of its own for the
    rank (`split_record`'s rule: its reverse-complement belongs at [L-b, L-a) of record r's result).
    Pieces of all ranks, in rank order, tile every record exactly once."""
    lengths = [int(x) for x in lengths]
    total = sum(lengths)
    bounds = [total * k // nparts for k in range(nparts + 1)]
    out = [[] for _ in range(nparts)]
    start = 0                      # plane position of the current record
    k = 0
    for r, L in enumerate(lengths):
        a = 0
        while a < L or (L == 0 and a == 0):
            if L == 0:
                out[min(k, nparts - 1)].append((r, 0, 0))
                break
            while k < nparts - 1 and start + a >= bounds[k + 1]:
                k += 1
            b = L
            if k < nparts - 1 and start + L > bounds[k + 1]:
                b = (bounds[k + 1] - start) // align * align
                if b <= a:         # the cut would fall at or before the piece's start: the next rank takes it from a
                    k += 1
                    continue
            out[k].append((r, a, b))
            a = b
        start += L
    return out


def allreduce_stats(partial, group=None):
    """partial: int64 tensor [num, len_sum, gap_sum, len_min, len_max] of this rank (layout of
    skh_stats_partial).  Returns the file-level totals on every rank: two small all-reduces
    (RCCL on GPUs, gloo on CPU)."""
    import torch
    import torch.distributed as dist
    sums = partial[:3].clone()
    ext = torch.stack([partial[4], -partial[3]])          # max(len_max), max(-len_min)
    dist.all_reduce(sums, op=dist.ReduceOp.SUM, group=group)
    dist.all_reduce(ext, op=dist.ReduceOp.MAX, group=group)
    return torch.stack([sums[0], sums[1], sums[2], -ext[1], ext[0]])
