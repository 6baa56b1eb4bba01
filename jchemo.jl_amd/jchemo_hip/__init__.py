"""jchemo_hip — MI355X-native drop-in for Jchemo.jl's plskern / plsnipals hot path (host-side mirror
of the reference interface over the C ABI in include/jchemo_hip.h)."""
from ._lib import Context, JchError, LIB_PATH, SYMBOLS, default_context, load, unique_id  # noqa: F401
from .plsr import (Lwplsr, LwplsrPred, Plsr, coef, lwplsr, lwplsr_predict, query_shard, colmajor_empty, ensure_mat, plskern, plskern_, plsnipals, plsnipals_, plssimp, plssimp_, plsrosa, plsrosa_, plswold, plswold_, predict,  # noqa: F401
                   summary, transform, vip, xfit, xresid, msep, rmsep, ssr, bias, r2, cor2, segmkf, segmts, gridscorelv, gridcvlv, mpar, Plsrda, dummy, plsrda, plsrda_predict, Plslda, plslda, plsqda, plslda_predict, Mbplsr, mbplsr, mbplsr_transform, mbplsr_predict)
