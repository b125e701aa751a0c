"""IFCB raw format: ``.adc`` (CSV; columns 15/16/17 = ROI width, height, start
byte, 0-based) + ``.roi`` (uint8 blob).  Mirrors the reference's
``sykepic/utils/ifcb.py`` (``sample_to_datetime`` :16, ``raw_to_numpy`` :121,
``next_roi`` :133) but never writes PNGs next to the raw data (quirk Q8):
ROIs go from the blob to arrays."""

import datetime
from pathlib import Path

import numpy as np


def sample_to_datetime(sample, isoformat=False):
    """``D20180703T093453_IFCB114`` -> aware UTC datetime (or ISO string)."""
    stamp = datetime.datetime.strptime(sample[1:16], "%Y%m%dT%H%M%S")
    stamp = stamp.replace(tzinfo=datetime.timezone.utc)
    return stamp.isoformat() if isoformat else stamp


def parse_adc_arrays(adc):
    """(roi numbers, widths, heights, start bytes) as int64 arrays, non-empty ROIs only - the columns `next_roi`
    reads (reference sykepic/utils/ifcb.py:133-145), parsed by pandas' C reader in one call instead of 20 k
    `line.split(",")` (a tenth of the per-sample host time of `sykepic prob`).  Anything the C reader does not turn
    into plain integers goes back to the line loop, which raises what the reference raises."""
    try:
        import pandas as pd
        t = pd.read_csv(adc, header=None, usecols=[15, 16, 17], dtype=str, engine="c", skip_blank_lines=False,
                        na_filter=False)
        w, h, start = (t[c].to_numpy() for c in (15, 16, 17))
        w, h, start = (np.array([int(v) for v in col], dtype=np.int64) if col.size < 64 else col.astype(np.int64)
                       for col in (w, h, start))
    except Exception:   # noqa: BLE001 - malformed file: the loop below reports it like the reference
        rows = parse_adc(adc)
        a = np.array(rows, dtype=np.int64).reshape(-1, 4)
        return a[:, 0], a[:, 1], a[:, 2], a[:, 3]
    num = np.arange(1, w.size + 1, dtype=np.int64)
    keep = (w >= 1) & (h >= 1)          # empty trigger: skipped, so ROI ids are sparse (Q11)
    return num[keep], w[keep], h[keep], start[keep]


def parse_adc(adc):
    """[(roi number (1-based line), width, height, start byte)] for non-empty ROIs."""
    rows = []
    with open(adc) as fh:
        for i, line in enumerate(fh, start=1):
            cols = line.split(",")
            w, h, start = int(cols[15]), int(cols[16]), int(cols[17])
            if w < 1 or h < 1:  # empty trigger: skipped, so ROI ids are sparse (Q11)
                continue
            rows.append((i, w, h, start))
    return rows


def read_rois(adc, roi):
    """[(roi number, HxW uint8 array)] — raises FileNotFoundError / ValueError
    on missing or truncated raw data like the reference does."""
    adc, roi = Path(adc), Path(roi)
    for f in (adc, roi):
        if not f.is_file():
            raise FileNotFoundError(f)
    blob = np.fromfile(roi, dtype=np.uint8)
    out = []
    for num, w, h, start in parse_adc(adc):
        out.append((num, blob[start:start + w * h].reshape((h, w))))  # ValueError if truncated
    return out


def raw_to_numpy(adc, roi):
    yield from read_rois(adc, roi)


def raw_to_png(adc, roi, out_dir=None, force=False):
    """Kept for ``sykepic train --save-images``-style tooling: writes
    ``<sample>_<roi:05>.png`` greyscale PNGs (PIL instead of cv2)."""
    from PIL import Image
    adc = Path(adc)
    sample = adc.with_suffix("").name
    out_dir = Path(adc.with_suffix("")) if not out_dir else Path(out_dir)
    out_dir.mkdir(parents=True, exist_ok=force)
    for num, img in read_rois(adc, roi):
        Image.fromarray(img).save(out_dir / f"{sample}_{num:05}.png")


def filter_out_quality_flagged_samples(sample_paths, exclusion_list):
    with open(exclusion_list) as fh:
        bad = [ln.strip() for ln in fh if ln.strip()]
    return [Path(p) for p in sample_paths if not any(b in str(p) for b in bad)]
