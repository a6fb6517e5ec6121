"""COCO run-length encoding of instance masks, with the run extraction on the device.

The reference's test loop encodes every pasted mask with pycocotools (mmdet/apis/test.py:59-67 ->
mmdet/core/mask/utils.py:36-63 ``encode_mask_results`` -> ``mask_util.encode``): each detection is a full-image bool
array (1 MB at 1024^2) that is copied to the host first -- 2 GB per image at 2000 detections.  Here the column-major run
boundaries are found on the GPU from the pasted masks and only the run lengths travel; the host side packs them into
pycocotools' compressed string (the LEB128-style code of cocoapi maskApi.c ``rleToString``: 5 data bits per character,
continuation bit 0x20, sign carried by bit 0x10, counts from the third on stored as differences to the count two places
back, offset 48).  pycocotools itself is not available in this image: the coder is checked by its inverse
(``rle_decode``) and against hand-derived strings of the published algorithm in tests/test_host_cpu.py.
"""
import numpy as np
import torch


def counts_to_string(counts):
    """cocoapi rleToString."""
    out = bytearray()
    for i, c in enumerate(counts):
        x = int(c)
        if i > 2:
            x -= int(counts[i - 2])
        more = True
        while more:
            ch = x & 0x1f
            x >>= 5
            more = (x != -1) if (ch & 0x10) else (x != 0)
            if more:
                ch |= 0x20
            out.append(ch + 48)
    return bytes(out)


def string_to_counts(s):
    """cocoapi rleFrString."""
    if isinstance(s, str):
        s = s.encode('ascii')
    counts, p = [], 0
    while p < len(s):
        x, k, more = 0, 0, True
        while more:
            ch = s[p] - 48
            x |= (ch & 0x1f) << (5 * k)
            more = bool(ch & 0x20)
            p += 1
            k += 1
            if not more and (ch & 0x10):
                x |= -1 << (5 * k)
        if len(counts) > 2:
            x += counts[-2]
        counts.append(x)
    return counts


def rle_decode(rle):
    """{'size': [h, w], 'counts': bytes} -> bool [h, w] (column-major runs, starting with a run of zeros)."""
    h, w = rle['size']
    flat = np.zeros(h * w, dtype=bool)
    pos, val = 0, False
    for c in string_to_counts(rle['counts']):
        if val:
            flat[pos:pos + c] = True
        pos += c
        val = not val
    return flat.reshape(w, h).T


@torch.no_grad()
def rle_encode_masks(masks, chunk=64):
    """masks: bool/uint8 device tensor [N, H, W] -> list of N RLE dicts ``{'size': [H, W], 'counts': bytes}``
    (what ``mask_util.encode(np.asfortranarray(mask))`` returns).  Run boundaries are computed on the device, ``chunk``
    masks at a time; only the boundary positions are copied to the host."""
    n, h, w = masks.shape
    out = []
    for s in range(0, n, chunk):
        m = masks[s:s + chunk].to(torch.bool).transpose(1, 2).reshape(-1, h * w)      # column-major flattening
        first = m[:, :1]
        change = torch.cat([first, m[:, 1:] != m[:, :-1]], 1)                            # a run starts here (mask value flips)
        idx = change.nonzero()                                                            # [(mask, position)], row-major sorted
        rows, pos = idx[:, 0].cpu().numpy(), idx[:, 1].cpu().numpy()
        split = np.searchsorted(rows, np.arange(m.shape[0] + 1))
        for i in range(m.shape[0]):
            p = pos[split[i]:split[i + 1]]
            bounds = np.concatenate([[0], p, [h * w]]).astype(np.int64)                   # leading run of zeros may be empty
            counts = np.diff(bounds)
            if len(p) and p[0] == 0:                 # mask starts with ones: the zero run has length 0 (already the first diff)
                pass
            out.append({'size': [h, w], 'counts': counts_to_string(counts.tolist())})
    return out


def encode_mask_results(segm_results):
    """Host-array form of ``encode_mask_results`` (utils.py:36-63) for results already on the host."""
    enc = [[] for _ in segm_results]
    for i, cls in enumerate(segm_results):
        if len(cls):
            arr = torch.from_numpy(np.stack(cls))
            enc[i] = rle_encode_masks(arr)
    return enc
