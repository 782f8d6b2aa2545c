"""STLSQ driver with the flag names of the reference's main_sindy.py / get_sindy_args.

The shipped reference script cannot run (it calls the shadowed Adam ``train_SINDy`` with the
STLSQ signature and lacks the noise/smoothing flags its dataset needs).  This driver
keeps the flag names and drives the sequential-threshold least-squares loop that *is* defined
(train.py:872-887): Gram on the GPU once, masked solves on the host.
"""
from __future__ import annotations

import numpy as np
import torch

from . import train as T
from .dataset import get_dataset
from .parser_utils import get_sindy_args
from .sindy import SINDyRegression


def main(argv=None):
    args = get_sindy_args(argv)
    torch.manual_seed(args.seed)
    np.random.seed(args.seed)
    args = vars(args)
    args.setdefault('noise', 0.0)
    args.setdefault('smoothing', None)
    train_dataset, _, args = get_dataset(args)
    args['L_list'] = []
    regressor = SINDyRegression(**args).to(args['device'])
    T.train_SINDy(regressor, train_dataset.x, train_dataset.dx, w_sindy_reg=args['w_reg'], **args)
    return regressor


if __name__ == '__main__':
    main()
