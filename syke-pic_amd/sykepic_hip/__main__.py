"""`python -m sykepic_hip` — the `sykepic` command line for the MI355X path.

Sub-commands and flags mirror the reference's ``sykepic/__main__.py`` for the
hot path: ``train`` (:43-61) and ``prob`` (:64-99).  ``predict`` is an alias
of ``prob`` (BASELINE.json names it so).  The pandas post-processing
sub-commands (class, feat, size, abundance, ...) are out of scope
 and are served by the reference itself from the CSVs written
here."""

from argparse import ArgumentParser


def build_parser():
    parser = ArgumentParser(prog="sykepic", description="CLI tool for plankton image classification at SYKE (MI355X hot path)")
    sub = parser.add_subparsers(title="available sub-commands", required=True, dest="sub-command",
                                help="sykepic {sub-command} -h for more information")
    tp = sub.add_parser("train", description="Train neural network classifiers")
    tp.add_argument("config", help="Path to config file")
    tp.add_argument("--collage", nargs=3, metavar=("ROWS", "COLUMNS", "PNG"),
                    help="Save a ROWS x COLUMNS grid of transformed images to PNG.")
    tp.add_argument("--dist", metavar="FILE", help="Save a class distribution plot to FILE")
    tp.add_argument("--save-images", metavar="DIR", help="Extract train, test, val images to this path")
    tp.set_defaults(func=_train)
    for name in ("prob", "predict"):
        pp = sub.add_parser(name, description="Calculate class probabilities")
        raw = pp.add_mutually_exclusive_group(required=True)
        raw.add_argument("-r", "--raw", metavar="DIR", help="Root directory of raw IFCB data")
        raw.add_argument("-s", "--samples", nargs="+", metavar="SAMPLE PATH",
                         help="One or more sample paths (raw file without suffix)")
        raw.add_argument("--image-dir", metavar="DIR", help="Root directory of images")
        raw.add_argument("--images", nargs="+", metavar="FILE", help="One or more image paths")
        pp.add_argument("-m", "--model", required=True, help="Model directory")
        pp.add_argument("-o", "--out", required=True, help="Root output directory")
        pp.add_argument("-b", "--batch-size", type=int, default=64, metavar="INT", help="Default is 64")
        pp.add_argument("-w", "--num-workers", type=int, default=2, metavar="INT", help="Default is 2")
        pp.add_argument("-f", "--force", action="store_true", help="Force overwrite of previous probabilities")
        pp.set_defaults(func=_prob)
    # not in the reference: one-off preparation of a model directory for the calibrated single-pass mode
    cp = sub.add_parser("calibrate", description="Measure activation means for a model directory (writes act_means.pth: "
                                                 "`prob` then runs every convolution as one fp16 product)")
    craw = cp.add_mutually_exclusive_group(required=True)
    craw.add_argument("-r", "--raw", metavar="DIR", help="Root directory of raw IFCB data")
    craw.add_argument("-s", "--samples", nargs="+", metavar="SAMPLE PATH", help="One or more sample paths")
    craw.add_argument("--image-dir", metavar="DIR", help="Root directory of images")
    craw.add_argument("--images", nargs="+", metavar="FILE", help="One or more image paths")
    cp.add_argument("-m", "--model", required=True, help="Model directory")
    cp.add_argument("-n", "--num-images", type=int, default=2048, metavar="INT", help="Images to measure on (default 2048)")
    cp.add_argument("-b", "--batch-size", type=int, default=64, metavar="INT", help="Default is 64")
    cp.set_defaults(func=_calibrate)
    return parser


def _calibrate(args):
    from . import prob
    return prob.calibrate_call(args)


def _train(args):
    from . import train
    return train.main(args)


def _prob(args):
    from . import prob
    return prob.call(args)


def main(argv=None):
    from . import logger
    logger.setup()
    args = build_parser().parse_args(argv)
    args.func(args)


if __name__ == "__main__":
    main()
