"""tools/tswgen/plan4.py -- the row-descriptor tables of the round-6 loop (kernel4.py): the same 16-byte descriptors and the same
plans as tools/tswgen/plan.py (band groups / the linear plan of the forward passes), with this loop's padding rows, table size and
last-step formula (12 waves x 3 rows: stream row q enters at step 2 (q div 3) + q mod 3).  Device version: cspn2d_tsw4.hip."""
import contextlib

from . import plan as P2
from . import kernel4 as K4


@contextlib.contextmanager
def _consts():
    old = (P2.PADF, P2.PADB, P2.TAB_MAX_ROWS)
    P2.PADF, P2.PADB, P2.TAB_MAX_ROWS = K4.PADF, K4.PADB, K4.TAB_MAX_ROWS
    try:
        yield
    finally:
        P2.PADF, P2.PADB, P2.TAB_MAX_ROWS = old


def _fix(hdr):
    for g in range(hdr.shape[0]):
        hdr[g, 1] = K4.last_step(int(hdr[g, 0]))
    return hdr


def build_plan(B, H, W, n_iter, n_wg, xcd=None):
    with _consts():
        hdr, tab = P2.build_plan(B, H, W, n_iter, n_wg, xcd)
    hdr[:, 2] = -1
    return _fix(hdr), tab


def build_plan_linear(B, H, W, n_iter, ncu, xcd=True):
    with _consts():
        lp, hdr, tab = P2.build_plan_linear(B, H, W, n_iter, ncu, xcd)
    return lp, _fix(hdr), tab
