"""Host logic of the prepared kNN-LWPLSR handle (jchemo_hip.plsr._lwplsr_prepared), with a stub in place of the shared
library (no GPU, no compute): the handle lives OUTSIDE the `Lwplsr` instance, so that copy / deepcopy / pickle of a
predicted-from model work, a copy never shares the original's handle, the handle is released exactly once when its object
dies, and re-assigning the model's data or metric rebuilds it."""
import copy
import ctypes as C
import gc
import pickle

import numpy as np

import jchemo_hip.plsr as P


class _StubLib:
    def __init__(self):
        self.next = 1000
        self.live = set()
        self.released = []

    def jch_lwplsr_prepare(self, ctx, loc, xa, n, p, ldx, ya, qk, ldy, za, ldz, dd, href):
        self.next += 1
        href._obj.value = self.next
        self.live.add(self.next)
        return 0

    def jch_lwplsr_add_query_map(self, *a):
        return 0

    def jch_lwplsr_release(self, ctx, h):
        v = h.value if hasattr(h, "value") else h
        assert v in self.live, "double release of a prepared handle"
        self.live.discard(v)
        self.released.append(v)
        return 0


class _StubCtx:
    _h = C.c_void_p(77)

    def check(self, status):
        assert status == 0


def _model():
    X = np.asfortranarray(np.arange(60.0).reshape(20, 3))
    Y = np.asfortranarray(np.arange(20.0).reshape(20, 1))
    return P.Lwplsr(X, Y, None, "eucl", 1.5, 5, 2, 1e-4, False)


def test_handle_is_kept_off_the_instance_and_copies_get_their_own(monkeypatch):
    stub = _StubLib()
    monkeypatch.setattr(P._lib, "load", lambda: stub)
    ctx = _StubCtx()
    obj = _model()
    st = P._lwplsr_prepared(obj, ctx, False, 1)
    assert P._lwplsr_prepared(obj, ctx, False, 1) is st                   # cached
    assert "_prep" not in obj.__dict__ and all(not isinstance(v, dict) for v in obj.__dict__.values())
    h0 = st["handle"].value
    # plain data: every copy protocol works after a predict
    c1, c2 = copy.copy(obj), copy.deepcopy(obj)
    c3 = pickle.loads(pickle.dumps(obj))
    for c in (c1, c2, c3):
        assert np.array_equal(c.X, obj.X) and c.metric == obj.metric
        hc = P._lwplsr_prepared(c, ctx, False, 1)["handle"].value
        assert hc != h0 and hc in stub.live                               # own handle, never the original's
    # the original dies: ITS handle is released once, the copies' handles stay valid
    del obj, st
    gc.collect()
    assert stub.released == [h0]
    for c in (c1, c2, c3):
        assert P._lwplsr_prepared(c, ctx, False, 1)["handle"].value in stub.live
    del c1, c2, c3, c
    gc.collect()
    assert not stub.live and len(stub.released) == 4


def test_changed_model_fields_rebuild_the_handle(monkeypatch):
    stub = _StubLib()
    monkeypatch.setattr(P._lib, "load", lambda: stub)
    ctx = _StubCtx()
    obj = _model()
    h0 = P._lwplsr_prepared(obj, ctx, False, 1)["handle"].value
    obj.X = np.asfortranarray(obj.X + 1.0)                                # new training data after the first predict
    h1 = P._lwplsr_prepared(obj, ctx, False, 1)["handle"].value
    assert h1 != h0 and stub.released == [h0]
    obj.scal = True                                                       # scal changes the neighbour space; plskern runs on the device,
    monkeypatch.setattr(P, "_knn_train_space", lambda o, c: (o.X, (lambda Xq: Xq), []))   # so stub the space builder here
    h2 = P._lwplsr_prepared(obj, ctx, False, 1)["handle"].value
    assert h2 != h1 and stub.released == [h0, h1]
    h3 = P._lwplsr_prepared(obj, ctx, False, 3)["handle"].value           # another response count handed to the batched kernel
    assert h3 != h2
    del obj
    gc.collect()
    assert not stub.live


def test_model_with_a_global_fit_is_collected(monkeypatch):
    """The query map of a model with a global PLS fit (nlvdis > 0) is a closure kept in the module-level handle table: it must hold
    the fit, not the `Lwplsr` object, or the object — and its device handle — would never be released."""
    stub = _StubLib()
    monkeypatch.setattr(P._lib, "load", lambda: stub)
    ctx = _StubCtx()
    X = np.asfortranarray(np.arange(60.0).reshape(20, 3))
    Y = np.asfortranarray(np.arange(20.0).reshape(20, 1))
    fm = P.Plsr(np.asfortranarray(np.ones((20, 2))), np.ones((3, 2)), np.ones((3, 2)), np.ones((3, 2)), np.ones((1, 2)), np.ones(2),
                np.zeros(3), np.ones(3), np.zeros(1), np.ones(1), np.full(20, 0.05), None)
    obj = P.Lwplsr(X, Y, fm, "eucl", 1.5, 5, 2, 1e-4, False)
    st = P._lwplsr_prepared(obj, ctx, False, 1)
    assert st["device_map"] and len(stub.live) == 1
    del obj, st
    gc.collect()
    assert not stub.live and len(stub.released) == 1
