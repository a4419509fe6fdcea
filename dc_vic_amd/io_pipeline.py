"""I/O front / back end of the encode-decode CLI: PNG decode and encode on worker threads,
pinned-memory staging, batches prefetched while the GPU codes the previous one.

The reference reads and writes one image at a time on the main thread (scripts/compress.py:52-70, 94, 131-135 and
src/utils/img_utils.py:19-44); at ~100 images/s per GPU that serial PNG work would be the bottleneck.  Pixel semantics
are the reference's: RGB, ToTensor (/255) then Normalize(.5, .5) on the way in; truncating uint8 (done on the GPU by
`crop_clamp`) on the way out.  PIL releases the GIL inside its codecs, so plain threads scale.
"""
from __future__ import annotations

import os
from concurrent.futures import Future, ThreadPoolExecutor
from typing import Callable, Iterable, Iterator, List, Optional, Sequence, Tuple

import numpy as np
import torch


def decode_png_u8(path: str) -> np.ndarray:
    """PNG -> uint8 HWC RGB (compress.py:62-66: Image.open(...).convert('RGB'))."""
    from PIL import Image
    with Image.open(path) as im:
        return np.asarray(im.convert("RGB"), dtype=np.uint8).copy()


def u8_to_model_range(u8_nhwc: torch.Tensor) -> torch.Tensor:
    """uint8 NHWC -> fp32 NCHW in [-1, 1]: ToTensor (x / 255) then Normalize(.5, .5) ((x - .5) / .5), same op order."""
    x = u8_nhwc.permute(0, 3, 1, 2).float().div(255.0)
    return (x - 0.5) / 0.5


def encode_png_u8(path: str, u8_hwc: np.ndarray) -> None:
    from PIL import Image
    Image.fromarray(np.ascontiguousarray(u8_hwc), mode="RGB").save(path)


def default_workers() -> int:
    """PNG worker threads of this rank: min(8, its share of the host cores) (affinity // LOCAL_WORLD_SIZE)."""
    from .parallel import host_core_budget
    return max(1, min(8, host_core_budget()))


class BatchPrefetcher:
    """Iterates over `chunks` (lists of image paths of one shape) yielding (chunk, x) with x fp32 NCHW in [-1, 1] on
    `device`.  `depth` batches ahead are being decoded on worker threads and staged through pinned memory; the
    host-to-device copy is asynchronous on the current stream."""

    def __init__(self, chunks: Sequence[Sequence[str]], device, workers: Optional[int] = None, depth: int = 2):
        self.chunks = list(chunks)
        self.device = torch.device(device)
        self.depth = max(1, depth)
        self.pool = ThreadPoolExecutor(max_workers=workers or default_workers(), thread_name_prefix="dcvic-png-dec")
        self._pin = self.device.type == "cuda"

    def _submit(self, chunk: Sequence[str]) -> List[Future]:
        return [self.pool.submit(decode_png_u8, p) for p in chunk]

    def _assemble(self, futs: List[Future]) -> torch.Tensor:
        arrs = [f.result() for f in futs]
        h, w = arrs[0].shape[:2]
        for a in arrs:
            if a.shape != (h, w, 3):
                raise ValueError("a batch needs equal image sizes (bucket the paths by shape first)")
        u8 = torch.empty((len(arrs), h, w, 3), dtype=torch.uint8, pin_memory=self._pin)
        for i, a in enumerate(arrs):
            u8[i] = torch.from_numpy(a)
        if self._pin:
            return u8_to_model_range(u8.to(self.device, non_blocking=True))    # uint8 over PCIe: 1/4 of the fp32 bytes
        return u8_to_model_range(u8)

    def __iter__(self) -> Iterator[Tuple[Sequence[str], torch.Tensor]]:
        pending: List[Tuple[Sequence[str], List[Future]]] = []
        it = iter(self.chunks)
        try:
            for _ in range(self.depth):
                c = next(it, None)
                if c is not None:
                    pending.append((c, self._submit(c)))
            while pending:
                chunk, futs = pending.pop(0)
                c = next(it, None)
                if c is not None:
                    pending.append((c, self._submit(c)))
                yield chunk, self._assemble(futs)
        finally:
            self.pool.shutdown(wait=True, cancel_futures=True)


class AsyncWriter:
    """Runs file writes (PNG encode, .bin containers) on worker threads; `close()` waits and re-raises the first error."""

    def __init__(self, workers: Optional[int] = None):
        self.pool = ThreadPoolExecutor(max_workers=workers or default_workers(), thread_name_prefix="dcvic-png-enc")
        self.futs: List[Future] = []

    def submit(self, fn: Callable, *args) -> None:
        self.futs.append(self.pool.submit(fn, *args))
        if len(self.futs) > 256:                      # bound the backlog (and the host memory it pins)
            self.futs.pop(0).result()

    def close(self) -> None:
        try:
            for f in self.futs:
                f.result()
        finally:
            self.futs = []
            self.pool.shutdown(wait=True)
