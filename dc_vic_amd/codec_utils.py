"""Bitstream container of the DC-VIC codec (byte-exact with src/utils/codec_utils.py:7-65).

header = uint16 H, uint16 W (little endian) + uint8 int(max|y_hat|) + uint8 quality  (6 bytes;
max_sample is written but never used by the decoder).  File = for each of
[header, z_string, y_string]: uint32 LE length + bytes.
"""
from __future__ import annotations

import struct
from typing import Dict, List, Tuple

import torch


class HeaderHandler:
    @staticmethod
    def check_img_size(img_size) -> None:
        assert len(img_size) == 2
        assert isinstance(img_size[0], int)
        assert isinstance(img_size[1], int)

    def encode(self, img_size: Tuple[int, int], y_hat, quality_ind: int) -> bytes:
        self.check_img_size(img_size)
        H, W = img_size
        if not (0 <= H < 65536 and 0 <= W < 65536):
            raise ValueError(f"image size {img_size} does not fit the uint16 header fields")
        max_val = int(torch.max(torch.abs(y_hat))) if isinstance(y_hat, torch.Tensor) else int(y_hat)
        # NumPy 1.24 (the reference's pin) wraps out-of-range values on uint8 conversion
        return struct.pack("<HH", H, W) + bytes([max_val & 0xFF]) + bytes([int(quality_ind) & 0xFF])

    def decode(self, header_byte_string: bytes) -> Dict:
        if len(header_byte_string) < 6:
            raise ValueError("header shorter than 6 bytes")
        H, W = struct.unpack("<HH", header_byte_string[:4])
        return {"img_size": (int(H), int(W)), "max_sample": int(header_byte_string[4]), "quality_ind": int(header_byte_string[5])}


def pack_byte_strings(string_list: List[bytes]) -> bytes:
    return b"".join(struct.pack("<I", len(s)) + s for s in string_list)


def unpack_byte_strings(blob: bytes) -> List[bytes]:
    out, off = [], 0
    while off < len(blob):
        if off + 4 > len(blob):
            raise ValueError("truncated length prefix")
        (n,) = struct.unpack("<I", blob[off:off + 4])
        if off + 4 + n > len(blob):
            raise ValueError("truncated string")
        out.append(blob[off + 4:off + 4 + n])
        off += 4 + n
    return out


def save_byte_strings(save_path: str, string_list: List[bytes]) -> None:
    with open(save_path, "wb") as f:
        f.write(pack_byte_strings(string_list))


def load_byte_strings(load_path: str) -> List[bytes]:
    with open(load_path, "rb") as f:
        return unpack_byte_strings(f.read())
