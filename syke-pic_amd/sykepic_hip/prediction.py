"""prob CSV -> (prediction, classified) per ROI with per-class thresholds.
Mirror of the reference's ``sykepic/compute/prediction.py``
(``prediction_dataframe`` :8, ``threshold_dictionary`` :31, ``row_prediction``
:49): the highest-probability class that clears ITS OWN threshold wins, else
arg-max with classified=False.  Vectorised over rows instead of a pandas
``apply``."""

from pathlib import Path

import numpy as np
import pandas as pd


def threshold_dictionary(thresholds, default=None):
    out = {}
    with open(thresholds) as fh:
        for line in fh:
            parts = line.strip().split()
            if not parts:
                continue
            if len(parts) > 1:
                out[parts[0]] = float(parts[1])
            elif default:
                out[parts[0]] = float(default)
            else:
                raise ValueError(f"Missing threshold for {parts[0]}, and no default value specified.")
    return out


def predict_arrays(probs, classes, thresholds):
    """probs [n, C] -> (index of predicted class [n], classified [n] bool)."""
    probs = np.asarray(probs, dtype=np.float64)
    top = probs.argmax(1)
    if isinstance(thresholds, (int, float)):
        return top, probs[np.arange(len(top)), top] > thresholds
    thr = np.array([thresholds.get(c, np.inf) for c in classes], dtype=np.float64)
    ok = probs >= thr[None, :]
    masked = np.where(ok, probs, -1.0)
    # stable descending order like Series.sort_values: first maximum wins
    best = masked.argmax(1)
    any_ok = ok.any(1)
    return np.where(any_ok, best, top), any_ok


def insert_prediction(df, thresholds):
    classes = list(df.columns)
    idx, ok = predict_arrays(df.to_numpy(), classes, thresholds)
    df.insert(0, "prediction", pd.Categorical([classes[i] for i in idx]))
    df.insert(1, "classified", ok.astype(bool))


def prediction_dataframe(probabilities, thresholds=0.0):
    if isinstance(probabilities, list):
        frames = []
        for csv in probabilities:
            df = pd.read_csv(csv)
            df.insert(0, "sample", Path(csv).with_suffix("").stem)
            df.set_index(["sample", "roi"], inplace=True)
            frames.append(df)
        df = pd.concat(frames)
    elif isinstance(probabilities, (str, Path)):
        df = pd.read_csv(probabilities, index_col=0)
    else:
        raise ValueError(f"Type {type(probabilities)} not allowed for probabilities")
    if isinstance(thresholds, (str, Path)):
        thresholds = threshold_dictionary(thresholds)
    if not df.empty:
        insert_prediction(df, thresholds)
    return df


def predict_gpu(probs, classes, thresholds):
    """GPU form of predict_arrays for a CUDA float32 [n, C] tensor
    (spk_predict_rows): returns (int32 class index [n], bool classified [n])
    tensors on the same device — fused after net_pass, no pandas apply."""
    import ctypes as C

    import torch

    from . import lib
    so = lib.load()
    probs = probs.contiguous().float()
    n, c = probs.shape
    pred = torch.empty(n, dtype=torch.int32, device=probs.device)
    ok = torch.empty(n, dtype=torch.uint8, device=probs.device)
    if isinstance(thresholds, (int, float)):
        thr, scalar = None, float(thresholds)
    else:
        thr = torch.tensor([thresholds.get(k, float("inf")) for k in classes], dtype=torch.float32,
                           device=probs.device)
        scalar = 0.0
    with torch.cuda.device(probs.device):
        stream = C.c_void_p(torch.cuda.current_stream(probs.device).cuda_stream)
        lib.check(so.spk_predict_rows(C.c_void_p(probs.data_ptr()), n, c,
                                      C.c_void_p(thr.data_ptr()) if thr is not None else None, scalar,
                                      C.c_void_p(pred.data_ptr()), C.c_void_p(ok.data_ptr()), stream))
    return pred, ok.bool()
