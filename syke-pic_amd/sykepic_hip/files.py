"""Output path convention and sample discovery (reference
``sykepic/utils/files.py``: ``sample_csv_path`` :27, ``list_sample_paths`` :40,
``list_sample_csvs`` :47)."""

from pathlib import Path

from . import ifcb


def sample_csv_path(sample_path, out_dir, suffix=None):
    """``<out>/YYYY/MM/DD/<sample><suffix>.csv`` from the sample's timestamp."""
    sample = Path(sample_path).name
    day = ifcb.sample_to_datetime(sample).strftime("%Y/%m/%d")
    return Path(out_dir) / day / f"{sample}{suffix or ''}.csv"


def list_sample_paths(root_dir, filter=None):
    paths = (roi.with_suffix("") for roi in Path(root_dir).glob("**/*.roi"))
    if filter is not None:
        paths = (p for p in paths if p.name in filter)
    return list(paths)


def list_sample_csvs(root_dir, filter=None):
    return [p for p in Path(root_dir).glob("**/*.csv") if not filter or p.with_suffix("").stem in filter]
