"""The one entry point that allocates nothing (include/mirge_hip.h: mirge_collapse_keys16_noalloc) against the entry
point it is a variant of, mirge_collapse_packed_dev: same distinct keys, same counts (bit-exact, compared as sets since
neither promises an order), on two calls in a row, the second with a smaller input, then the size limit."""
import numpy as np
import pytest

from mirge_amd import capi, synth

pytestmark = pytest.mark.gpu


def _keys16(ctx, torch, n, seed):
    libs, _ = synth.make_libraries(synth.Profile.tiny(), synth.BASE_SEED + 11)
    seqs, lens = synth.make_reads(libs, n, seed)
    d_seqs = torch.from_numpy(seqs).cuda(); d_lens = torch.from_numpy(lens).cuda()
    k16 = torch.zeros((n, 2), dtype=torch.int64, device="cuda"); k48 = torch.zeros((n, 6), dtype=torch.int64, device="cuda")
    n16, _ = ctx.pack_dev(d_seqs.data_ptr(), seqs.shape[1], d_lens.data_ptr(), n, k16.data_ptr(), k48.data_ptr())
    return k16[:n16].contiguous()


def _table_of(keys_u64, counts):
    return {(int(a), int(b)): int(c) for (a, b), c in zip(keys_u64.tolist(), counts.tolist())}


def test_no_alloc_collapse_equals_collapse_packed_dev():
    import torch
    with capi.Context(None, 1, 0) as ctx:
        uk = torch.empty((300_000, 2), dtype=torch.int64, device="cuda")
        uc = torch.empty(300_000, dtype=torch.int32, device="cuda")
        empty48 = torch.zeros((1, 6), dtype=torch.int64, device="cuda")
        for n, seed in ((300_000, 5), (70_001, 6)):           # second call smaller: the array still holds the first call's state
            k16 = _keys16(ctx, torch, n, synth.BASE_SEED + seed)
            m = k16.shape[0]
            uk.fill_(-1); uc.fill_(-1)
            torch.cuda.synchronize()
            free_before = torch.cuda.mem_get_info()[0]          # hipMemGetInfo: what the device has left
            U = ctx.collapse_keys16_noalloc(k16.data_ptr(), m, uk.data_ptr(), uc.data_ptr())
            assert torch.cuda.mem_get_info()[0] == free_before, "the call took (or gave back) device memory"
            got = _table_of(uk[:U].cpu().numpy().view(np.uint64), uc[:U].cpu().numpy().view(np.uint32))
            assert len(got) == U                                # distinct keys came out distinct
            ctx.reset()
            ctx.collapse_packed_dev(k16.data_ptr(), m, empty48.data_ptr(), 0, 0)
            assert ctx.n_unique() == U
            useqs, ulens, _, _, _ = ctx.export_uniques()
            uid, smp, cnt = ctx.export_counts()
            # the existing entry point hands sequences back: pack them the way keys.h does to compare key for key
            code = np.zeros(256, dtype=np.uint64); code[ord("C")] = 1; code[ord("G")] = 2; code[ord("T")] = 3
            w0 = np.zeros(U, dtype=np.uint64); w1 = np.zeros(U, dtype=np.uint64)
            for i in range(56):
                cc = np.where(i < ulens, code[useqs[:, i]], 0).astype(np.uint64)
                if i < 32:
                    w0 |= cc << np.uint64(2 * i)
                else:
                    w1 |= cc << np.uint64(2 * (i - 32))
            w1 |= ulens.astype(np.uint64) << np.uint64(56)
            percount = np.zeros(U, dtype=np.uint64); np.add.at(percount, uid, cnt)
            exp = {(int(a), int(b)): int(c) for a, b, c in zip(w0.tolist(), w1.tolist(), percount.tolist())}
            assert got == exp
            assert sum(got.values()) == m
        # untouched beyond U, limit enforced, empty input fine
        assert int(uc[U:].max()) == -1
        big = capi.NOALLOC_MAX_KEYS + 1
        with pytest.raises(capi.MirgeError) as ei:
            ctx.collapse_keys16_noalloc(k16.data_ptr(), big, uk.data_ptr(), uc.data_ptr())
        assert ei.value.code == capi.MIRGE_E_LIMIT
        assert ctx.collapse_keys16_noalloc(k16.data_ptr(), 0, uk.data_ptr(), uc.data_ptr()) == 0


def test_no_alloc_collapse_at_its_maximum_size():
    """n = MIRGE_NOALLOC_MAX_KEYS, nearly all keys distinct: the slot table at its largest, the per-key counters and ranks
    filled to their last element; against mirge_collapse_packed_dev, and without a byte of device memory taken."""
    import torch
    n = capi.NOALLOC_MAX_KEYS
    rng = np.random.default_rng(77)
    w0 = rng.integers(0, 1 << 62, size=n, dtype=np.int64) & ((1 << 62) - 1)
    w0[: n // 50] = w0[n // 50: 2 * (n // 50)]                  # 2 % duplicates
    keys = np.stack([w0, np.full(n, 31 << 56, dtype=np.int64)], axis=1)      # 31-nt reads: all bits of w0 below 2 * 31 are sequence
    with capi.Context(None, 1, 0) as ctx:
        k16 = torch.from_numpy(keys).cuda()
        uk = torch.empty((n, 2), dtype=torch.int64, device="cuda"); uc = torch.empty(n, dtype=torch.int32, device="cuda")
        empty48 = torch.zeros((1, 6), dtype=torch.int64, device="cuda")
        ctx.validate_keys_dev(k16.data_ptr(), n, empty48.data_ptr(), 0)
        torch.cuda.synchronize()
        free_before = torch.cuda.mem_get_info()[0]
        U = ctx.collapse_keys16_noalloc(k16.data_ptr(), n, uk.data_ptr(), uc.data_ptr())
        assert torch.cuda.mem_get_info()[0] == free_before
        got = _table_of(uk[:U].cpu().numpy().view(np.uint64), uc[:U].cpu().numpy().view(np.uint32))
        uniq, cnt = np.unique(keys[:, 0].view(np.uint64), return_counts=True)
        assert U == len(uniq) == len(got)
        assert got == {(int(a), 31 << 56): int(c) for a, c in zip(uniq.tolist(), cnt.tolist())}
        ctx.collapse_packed_dev(k16.data_ptr(), n, empty48.data_ptr(), 0, 0)
        assert ctx.n_unique() == U
