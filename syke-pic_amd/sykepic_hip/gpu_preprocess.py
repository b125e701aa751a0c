"""ROI preprocessing on the GPU straight from the `.roi` bytes
(`spk_preprocess_rois`): replaces, for the standard eval pipeline
(Resize + ToTensor, 3 channels), the reference's PNG round trip and
per-image host transform (sykepic/compute/probability.py:143-155)."""

import ctypes as C

import numpy as np
import torch

from . import ifcb, lib, preprocess


def supported(transform, num_chans):
    """GPU path handles exactly the eval pipeline the reference builds."""
    if num_chans != 3 or not isinstance(transform, preprocess.Compose):
        return False
    kinds = [type(t) for t in transform.transforms]
    if kinds != [preprocess.Resize, preprocess.ToTensor]:
        return False
    th, tw = transform.target_dims
    return transform.border in ("mode", (0, 0, 0), (255, 255, 255)) and (th * tw * 3) % 4 == 0


def border_code(transform):
    return -1 if transform.border == "mode" else int(transform.border[0])


class SampleOnGpu:
    """The `.roi` blob of one sample in device memory + its ROI table."""

    def __init__(self, adc, roi, device):
        num, w, h, start = ifcb.parse_adc_arrays(adc)
        blob = np.fromfile(roi, dtype=np.uint8)
        over = start + w * h > blob.size
        if over.any():
            raise ValueError(f"ROI {int(num[np.argmax(over)])} exceeds the .roi file")
        self.numbers = num                      # int64 array, ascending
        self.device = torch.device(device)
        self.blob = torch.from_numpy(blob).to(self.device) if blob.size else torch.zeros(1, dtype=torch.uint8, device=self.device)
        rois = np.zeros(num.size, dtype=np.dtype([("offset", "<i8"), ("width", "<i4"), ("height", "<i4")]))
        rois["offset"], rois["width"], rois["height"] = start, w, h
        self.rois = torch.from_numpy(rois.view(np.uint8).copy()).to(self.device) if num.size else None
        self.blob_bytes = int(blob.size)

    def __len__(self):
        return len(self.numbers)

    def batch(self, begin, end, out_h, out_w, border):
        """uint8 [end-begin, out_h, out_w, 3] on the GPU."""
        n = end - begin
        out = torch.empty((n, out_h, out_w, 3), dtype=torch.uint8, device=self.device)
        so = lib.load()
        with torch.cuda.device(self.device):
            stream = C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)
            lib.check(so.spk_preprocess_rois(C.c_void_p(self.blob.data_ptr()), self.blob_bytes,
                                             C.c_void_p(self.rois.data_ptr() + begin * C.sizeof(lib.Roi)), n,
                                             out_h, out_w, border, C.c_void_p(out.data_ptr()), stream))
        return out
