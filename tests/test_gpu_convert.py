"""GPU parity of FASTQ quality re-encoding: skh_quality_table against the oracle's
ora_quality_table for every (from, to, force), and the two device steps `convert` is made of
(skh_validate_dev with the validity table, skh_revcomp_dev with the conversion table and no
reversal) against the oracle's tables applied byte by byte.  The oracle is pinned by the
documented examples of doc/docs/usage.md:878-956."""
import ctypes as C

import numpy as np
import pytest

from helpers import make_batch, ptr, to_dev, u8p

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def env(oracle_lib):
    import torch
    import seqkit_amd
    from seqkit_amd import capi
    lib = seqkit_amd.load()
    ctx = lib.init(0)
    oracle_lib.ora_quality_table.restype = C.c_int
    oracle_lib.ora_quality_table.argtypes = [C.c_int, C.c_int, C.c_int, u8p, u8p]
    yield {"lib": lib, "ctx": ctx, "torch": torch, "capi": capi, "ora": oracle_lib}
    lib.skh_destroy(ctx)


def tables(env, frm, to, force):
    lut, valid = (C.c_uint8 * 256)(), (C.c_uint8 * 256)()
    ident = C.c_int(0)
    env["lib"].check(env["lib"].skh_quality_table(frm, to, force, lut, valid, C.byref(ident)))
    wl, wv = np.zeros(256, np.uint8), np.zeros(256, np.uint8)
    wi = env["ora"].ora_quality_table(frm, to, force, wl.ctypes.data_as(u8p), wv.ctypes.data_as(u8p))
    return lut, valid, ident.value, wl, wv, wi


@pytest.mark.parametrize("frm", [1, 2, 3, 4, 5])
@pytest.mark.parametrize("to", [1, 2, 3, 4, 5])
@pytest.mark.parametrize("force", [0, 1])
def test_quality_tables_and_planes(env, frm, to, force):
    torch, lib = env["torch"], env["lib"]
    lut, valid, ident, wl, wv, wi = tables(env, frm, to, force)
    assert ident == wi
    assert np.array_equal(np.frombuffer(bytes(lut), np.uint8), wl)
    assert np.array_equal(np.frombuffer(bytes(valid), np.uint8), wv)
    # a quality plane of valid bytes only, plus one planted invalid byte when the pair has any
    rng = np.random.default_rng(frm * 10 + to)
    ok = np.nonzero(wv)[0].astype(np.uint8)
    ok = ok[(ok >= 33) & (ok <= 126)]
    lens = np.concatenate([rng.integers(0, 300, 4000), [0, 1, 16, 17, 40000]])
    qual, _, offs = make_batch(lens, seed=7, letters=ok.tobytes())
    total = int(offs[-1])
    d_q, d_offs = to_dev(qual), to_dev(offs)
    d_out = torch.zeros_like(d_q)
    d_res = torch.zeros(3, dtype=torch.int64, device="cuda:0")
    torch.cuda.synchronize()
    lib.check(lib.skh_validate_dev(env["ctx"], ptr(d_q), ptr(d_offs), len(lens), total, valid, 0, ptr(d_res), None))
    lib.check(lib.skh_revcomp_dev(env["ctx"], ptr(d_q), ptr(d_out), ptr(d_offs), len(lens), total, lut, 0, None))
    lib.check(lib.skh_sync(env["ctx"]))
    assert env["capi"].ValidateResult.from_buffer_copy(d_res.cpu().numpy().tobytes()).found == 0
    assert np.array_equal(d_out.cpu().numpy()[:total], wl[qual[:total]])
    bad = np.nonzero((wv == 0) & (np.arange(256) >= 33) & (np.arange(256) <= 126))[0]
    if len(bad):
        rec = 1234
        pos = int(offs[rec]) + int(lens[rec]) - 1 if lens[rec] else int(offs[rec + 1])
        q2 = qual.copy(); q2[pos] = bad[-1]; q2[int(offs[3000]) + 1 if lens[3000] > 1 else pos] = bad[0]
        d_q2 = to_dev(q2)
        torch.cuda.synchronize()
        lib.check(lib.skh_validate_dev(env["ctx"], ptr(d_q2), ptr(d_offs), len(lens), total, valid, 0, ptr(d_res), None))
        lib.check(lib.skh_sync(env["ctx"]))
        r = env["capi"].ValidateResult.from_buffer_copy(d_res.cpu().numpy().tobytes())
        first = int(np.nonzero(wv[q2[:total]] == 0)[0][0])
        want_rec = int(np.searchsorted(offs, first, side="right")) - 1
        assert (r.found, r.record, r.offset, r.letter) == (1, want_rec, first - int(offs[want_rec]), int(q2[first]))
