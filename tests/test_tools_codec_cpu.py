"""Host code of the product without a GPU: key generation, bit encryption,
the file codec and that libieache.so exports every symbol
include/ieache.h declares."""
import ctypes
import os
import re

import numpy as np
import pytest

from np_tfhe import negacyclic_mul_binary, _wrap32

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_c_abi_exports_every_declared_symbol(ia):
    hdr = open(os.path.join(ROOT, "include", "ieache.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    names = set(re.findall(r"\b(ieache_[a-z_0-9]+)\s*\(", hdr))
    assert len(names) >= 30
    L = ctypes.CDLL(ia.library_path())
    missing = [n for n in sorted(names) if not hasattr(L, n)]
    assert not missing, missing
    assert b"gfx950" in L.ieache_version() if not isinstance(L.ieache_version(), int) else True


def test_circuit_info_entry_point_of_header_0_1_writes_its_own_struct_only(ia):
    """ieache_circuit_info grew from 56 to 72 bytes after header version 0.1; ieache_circuit_info_get() existed then, so it
    fills the 56-byte prefix and leaves what follows in the caller's memory alone.  _get_ex fills the current struct."""
    L = ia.lib()
    buf = (ctypes.c_ubyte * 96)(*([0xAB] * 96))
    L.ieache_circuit_info_get.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
    assert L.ieache_circuit_info_get(ia.CIRC_MUL, 32, buf) == 0
    L.ieache_circuit_info_get.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.POINTER(ia.CircuitInfo)]
    raw = bytes(buf)
    assert raw[56:] == b"\xab" * 40
    full = ia.circuit_info(ia.CIRC_MUL, 32)
    assert ctypes.sizeof(ia.CircuitInfo) == 72 and raw[:56] == bytes(full)[:56]
    assert full.bootstraps == 11264 == full.reference_bootstraps and full.sched_levels == full.depth == 255


def test_default_params_are_libtfhe_128bit(ia):
    p = ia.default_params()
    assert (p.n, p.N, p.k, p.l, p.Bgbit, p.ks_t, p.ks_basebit) == (630, 1024, 1, 3, 7, 8, 2)
    assert p.lwe_alpha_min == 2.0 ** -15 and p.tlwe_alpha_min == 2.0 ** -25
    assert p.bk_count * 4 == 30965760 and p.ksk_count * 4 == 82706432
    assert 4 * p.n + 16 == 2536 and 64 * 2536 == 162304                # dragonfly_cipher_cloud.py:1295


def test_encrypt_decrypt_roundtrip(ia, make_keys):
    kb = make_keys(12, 64)
    bits = np.random.default_rng(0).integers(0, 2, size=500).astype(np.uint8)
    ct = kb.enc(bits, 42)
    assert np.array_equal(kb.dec(ct), bits)
    # phases sit at +-1/8 with sigma 2^-15 noise
    ph = (ct[:, -1].astype(np.int64) - (ct[:, :-1].astype(np.int64) * kb.lwe_key).sum(1)) & 0xFFFFFFFF
    ph = np.where(ph >= 2 ** 31, ph - 2 ** 32, ph)
    assert np.all(np.abs(np.abs(ph) - 2 ** 29) < 2 ** 32 * 2.0 ** -15 * 8)
    assert np.array_equal(kb.enc(bits, 42), ct)       # deterministic per seed
    assert not np.array_equal(kb.enc(bits, 43), ct)


def test_keygen_structure(ia, make_keys):
    """BK_i rows are TLWE encryptions of s_i * gadget; KSK rows encrypt d*s'_i/base^(j+1)."""
    kb = make_keys(5, 64)
    p = kb.p
    assert set(np.unique(kb.lwe_key)) <= {0, 1} and set(np.unique(kb.tlwe_key)) <= {0, 1}
    tol = 2 ** 32 * p.tlwe_alpha_min * 8
    for i in range(p.n):
        for bloc in range(2):
            for q in range(p.l):
                row = kb.bk[i, bloc * p.l + q]
                phase = _wrap32(row[1].astype(np.int64) - negacyclic_mul_binary(row[0], kb.tlwe_key))
                h = int(kb.lwe_key[i]) << (32 - (q + 1) * p.Bgbit)
                expect = np.zeros(p.N, dtype=np.int64)
                if bloc == 1:
                    expect[0] = h
                else:  # gadget on the mask poly: phase = -h * s(X)
                    expect = -h * kb.tlwe_key.astype(np.int64)
                diff = _wrap32(phase.astype(np.int64) - expect).astype(np.int64)
                assert np.abs(diff).max() < tol
    tol = 2 ** 32 * p.lwe_alpha_min * 8
    base = 1 << p.ks_basebit
    for i in range(0, p.N, 7):
        for j in range(p.ks_t):
            assert not kb.ksk[i, j, 0].any()
            for d in range(1, base):
                s = kb.ksk[i, j, d]
                phase = int(s[-1]) - int((s[:-1].astype(np.int64) * kb.lwe_key).sum())
                msg = int(kb.tlwe_key[i]) * d * (1 << (32 - (j + 1) * p.ks_basebit))
                diff = int(_wrap32(phase - msg))
                assert abs(diff) < tol
    # deterministic per seed, different across seeds
    from ieache_amd import tools
    again = tools.keygen_raw(p, (1, 2, 3))
    assert np.array_equal(again["bk"], kb.bk) and np.array_equal(again["ksk"], kb.ksk)
    other = tools.keygen_raw(p, (1, 2, 4))
    assert not np.array_equal(other["lwe_key"], kb.lwe_key) or not np.array_equal(other["bk"], kb.bk)


def test_key_files_roundtrip_and_sizes(ia, tmp_path):
    from ieache_amd import tools
    p = ia.default_params().copy(n=9, N=32)
    tools.keygen_files(tmp_path, p, seed=(314, 1592, 657), nbit_seed=(314, 1592, 888))
    for name in ("secret.key", "cloud.key", "nbit.key"):  # keygen.c:38-50
        assert (tmp_path / name).exists()
    pc, bk, ksk = tools.read_cloud_key(tmp_path / "cloud.key")
    ps, lwe, tlwe = tools.read_secret_key(tmp_path / "secret.key")
    ref = tools.keygen_raw(p, (314, 1592, 657))
    assert np.array_equal(bk, ref["bk"].ravel()) and np.array_equal(ksk, ref["ksk"].ravel())
    assert np.array_equal(lwe, ref["lwe_key"]) and np.array_equal(tlwe, ref["tlwe_key"])
    for q in (pc, ps):
        assert bytes(q) == bytes(p)
    pn, nlwe, _ = tools.read_secret_key(tmp_path / "nbit.key")
    assert not np.array_equal(nlwe, lwe)  # different seed (keygen.c:34)
    raw = (tmp_path / "cloud.key").read_bytes()
    assert raw.startswith(b"-----BEGIN GATEBOOTSPARAMS-----\n")
    # libtfhe's section order as best known: write_tGswParams emits its TLWEPARAMS first
    order = [raw.index(b"-----BEGIN %s-----" % t) for t in (b"GATEBOOTSPARAMS", b"LWEPARAMS", b"TLWEPARAMS", b"TGSWPARAMS")]
    assert order == sorted(order)
    hdr_end = raw.index(b"-----END TGSWPARAMS-----\n") + len(b"-----END TGSWPARAMS-----\n")
    # body: tag, KS tag, variance, KS coefficients (all `base` entries), variance, BK coefficients
    assert len(raw) - hdr_end == 4 + 4 + 8 + p.ksk_count * 4 + 8 + p.bk_count * 4
    sec = (tmp_path / "secret.key").stat().st_size
    assert sec == len(raw) + 4 + 4 * p.n + 4 + 4 * p.N
    # truncated / corrupted files are refused, not misread
    (tmp_path / "bad.key").write_bytes(raw[:-10])
    with pytest.raises(ia.IeacheError):
        tools.read_cloud_key(tmp_path / "bad.key")
    (tmp_path / "bad2.key").write_bytes(raw.replace(b"TGSWPARAMS", b"XGSWPARAMS"))
    with pytest.raises(ia.IeacheError):
        tools.read_cloud_key(tmp_path / "bad2.key")
    with pytest.raises(ia.IeacheError):
        tools.read_cloud_key(tmp_path / "missing.key")


def _sections(raw):
    """split a key file written by this build into its four text sections and the binary rest"""
    secs = {}
    for t in (b"GATEBOOTSPARAMS", b"LWEPARAMS", b"TLWEPARAMS", b"TGSWPARAMS"):
        a = raw.index(b"-----BEGIN %s-----\n" % t)
        e = b"-----END %s-----\n" % t
        b = raw.index(e) + len(e)
        secs[t] = raw[a:b]
    body = raw[raw.index(b"-----END TGSWPARAMS-----\n") + len(b"-----END TGSWPARAMS-----\n"):]
    return secs, body


def test_key_reader_tolerates_other_layouts(ia, tmp_path):
    """SURVEY App. B codec strategy: no libtfhe-written file exists here, so the reader finds text
    sections by title in any order and solves for the binary layout among enumerated hypotheses.
    Every variant below carries the same key and must load to the same arrays."""
    import struct
    from ieache_amd import tools
    L = ia.lib()
    p = ia.default_params().copy(n=7, N=32)
    tools.keygen_files(tmp_path, p, seed=(5, 6, 7), nbit_seed=(8, 9))
    ref = tools.keygen_raw(p, (5, 6, 7))
    bk, ksk = ref["bk"].ravel(), ref["ksk"].ravel()
    raw = (tmp_path / "cloud.key").read_bytes()
    secs, body = _sections(raw)
    tools.read_cloud_key(tmp_path / "cloud.key")
    assert b"KS{tag, one variance, all base rows} then BK{one variance}" in L.ieache_last_key_layout()
    G, LW, TL, TG = (secs[t] for t in (b"GATEBOOTSPARAMS", b"LWEPARAMS", b"TLWEPARAMS", b"TGSWPARAMS"))
    i32 = lambda v: struct.pack("<i", v)
    f64 = lambda v: struct.pack("<d", v)
    S = p.n + 1
    ksk4 = ksk.reshape(-1, 4, S)  # [N*t][base][n+1]
    ks_lwe = b"-----BEGIN LWEKSPARAMS-----\nbasebit: 2\nn: 32\nt: 8\n-----END LWEKSPARAMS-----\n"
    variants = {
        # this build's earlier header order (TGSW before TLWE) and a shuffled one
        "old_order": (G + LW + TG + TL + body, b"KS{tag, one variance"),
        "shuffled": (TL + G + TG + LW + body, b"KS{tag, one variance"),
        # property lines reordered inside a section, CRLF-free variants keep working
        "props": (G + b"-----BEGIN LWEPARAMS-----\nn: 7\nalpha_min: %.17g\nalpha_max: %.17g\n-----END LWEPARAMS-----\n"
                  % (p.lwe_alpha_min, p.lwe_alpha_max) + TL + TG + body, b"KS{tag"),
        # no type tags, no variances
        "bare": (G + LW + TL + TG + ksk.tobytes() + bk.tobytes(), b"KS{no variance, all base rows} then BK{no variance}"),
        # bootstrapping key first, key-switch key with its own text section in between (write_lweKeySwitchKey)
        "bk_first": (G + LW + TL + TG + i32(201) + f64(-1.0) + bk.tobytes() + ks_lwe + i32(200) + f64(2.0 ** -30) + ksk.tobytes(),
                     b"BK{one variance} then KS{tag, one variance, all base rows}"),
        # the never-read d = 0 rows left out
        "no_d0": (G + LW + TL + TG + i32(201) + i32(200) + f64(0.0) + np.ascontiguousarray(ksk4[:, 1:]).tobytes() + f64(0.0) + bk.tobytes(),
                  b"d=0 rows omitted"),
        # a variance double after every key-switch sample and every TLWE row
        "per_sample": (G + LW + TL + TG + i32(201) + i32(200)
                       + b"".join(r.tobytes() + f64(1e-9) for r in ksk.reshape(-1, S))
                       + b"".join(r.tobytes() + f64(1e-15) for r in bk.reshape(-1, 2 * p.N)),
                       b"KS{tag, variance per sample, all base rows} then BK{variance per sample}"),
    }
    for name, (blob, want) in variants.items():
        f = tmp_path / (name + ".key")
        f.write_bytes(blob)
        q, b2, k2 = tools.read_cloud_key(f)
        assert bytes(q) == bytes(p), name
        assert np.array_equal(b2, bk), name
        got = k2.reshape(-1, 4, S)
        assert np.array_equal(got[:, 1:], ksk4[:, 1:]) and not got[:, 0].any(), name
        assert want in L.ieache_last_key_layout(), (name, L.ieache_last_key_layout())
    # a key whose d = 0 rows are NOT zero (a generator that encrypts h = 0 with noise like every other row): lweKeySwitch never
    # reads those rows, so the key is as good as any and must load -- the all-zero property only breaks ties (below)
    noisy = ksk4.copy()
    noisy[:, 0] = np.random.default_rng(11).integers(-2 ** 31, 2 ** 31, size=noisy[:, 0].shape, dtype=np.int64).astype(np.int32)
    assert ksk.tobytes() in body
    f = tmp_path / "noisy_d0.key"
    f.write_bytes(G + LW + TL + TG + body.replace(ksk.tobytes(), noisy.tobytes()))
    q, b2, k2 = tools.read_cloud_key(f)
    assert np.array_equal(b2, bk) and np.array_equal(k2.reshape(-1, 4, S)[:, 1:], ksk4[:, 1:]), L.ieache_last_key_layout()
    # ... without tags or variances nothing is left to tell [KSK][BK] from [BK][KSK]: decoded right or refused, never shifted
    f = tmp_path / "noisy_d0_bare.key"
    f.write_bytes(G + LW + TL + TG + noisy.tobytes() + bk.tobytes())
    try:
        q, b2, k2 = tools.read_cloud_key(f)
        assert np.array_equal(b2, bk) and np.array_equal(k2.reshape(-1, 4, S)[:, 1:], ksk4[:, 1:])
    except ia.IeacheError as e:
        assert "ambiguous key layout" in str(e)
    # layouts of EQUAL size that differ only in where a variance double stands (no tags to tell them apart): the first
    # eight bytes of a key-switch key that carries its d = 0 rows are zero and read as a perfectly plausible variance
    # 0.0, so [KSK][var][BK] also "fits" [var][KSK][BK] -- shifted by eight bytes.  The all-zero d = 0 rows decide.
    for name, blob in (("var_after_ksk", G + LW + TL + TG + ksk.tobytes() + f64(2.0 ** -30) + bk.tobytes()),
                       ("var_before_ksk", G + LW + TL + TG + f64(0.0) + ksk.tobytes() + bk.tobytes())):
        f = tmp_path / (name + ".key")
        f.write_bytes(blob)
        q, b2, k2 = tools.read_cloud_key(f)
        assert np.array_equal(b2, bk) and np.array_equal(k2, ksk), (name, L.ieache_last_key_layout())
    # ... and where the structure cannot decide (here: d = 0 rows omitted, so nothing is known to be zero, and the eight
    # bytes in question are plausible either way) the reader refuses instead of guessing
    amb = G + LW + TL + TG + f64(0.0) + np.ascontiguousarray(ksk4[:, 1:]).tobytes() + bk.tobytes()
    (tmp_path / "amb.key").write_bytes(amb)
    try:
        q, b2, k2 = tools.read_cloud_key(tmp_path / "amb.key")
        assert np.array_equal(b2, bk) and np.array_equal(k2.reshape(-1, 4, S)[:, 1:], ksk4[:, 1:])  # decoded right ...
    except ia.IeacheError as e:
        assert "ambiguous key layout" in str(e)                                                    # ... or refused, never shifted
    # secret key sets: keys before the cloud body, untagged, TGSW key first
    sraw = (tmp_path / "secret.key").read_bytes()
    _, sbody = _sections(sraw)
    cloud_body = sbody[:len(body)]
    lwe, tlwe = ref["lwe_key"], ref["tlwe_key"]
    hdr = G + LW + TL + TG
    svariants = {
        "keys_first": hdr + i32(43) + lwe.tobytes() + i32(202) + tlwe.tobytes() + cloud_body,
        "untagged": hdr + cloud_body + lwe.tobytes() + tlwe.tobytes(),
        "tgsw_first": hdr + cloud_body + i32(202) + tlwe.tobytes() + i32(43) + lwe.tobytes(),
    }
    for name, blob in svariants.items():
        f = tmp_path / (name + ".skey")
        f.write_bytes(blob)
        q, l2, t2 = tools.read_secret_key(f)
        assert np.array_equal(l2, lwe) and np.array_equal(t2, tlwe), name
    # nothing fits: the message says how many bytes were left and for which parameters
    (tmp_path / "odd.key").write_bytes(raw + b"\0\0\0")
    with pytest.raises(ia.IeacheError, match="no key layout fits"):
        tools.read_cloud_key(tmp_path / "odd.key")
    (tmp_path / "tag.key").write_bytes(hdr + i32(201) + i32(999) + body[8:])
    with pytest.raises(ia.IeacheError, match="no key layout fits"):
        tools.read_cloud_key(tmp_path / "tag.key")
    (tmp_path / "ks.key").write_bytes(hdr + ks_lwe.replace(b"t: 8", b"t: 5") + body)
    with pytest.raises(ia.IeacheError, match="LWEKSPARAMS"):
        tools.read_cloud_key(tmp_path / "ks.key")


def test_sample_stream_layout(ia, tmp_path):
    """One LweSample = int32 42 | a[n] | b | double: 4n+16 bytes (2536 at n=630)."""
    from ieache_amd import tools
    n = 630
    rows = np.random.default_rng(0).integers(-2 ** 31, 2 ** 31, size=(64, n + 1), dtype=np.int64).astype(np.int32)
    f = tmp_path / "answer.data"
    tools.write_samples(f, rows)
    assert f.stat().st_size == 162304  # the reference's failure marker size
    raw = f.read_bytes()
    assert int.from_bytes(raw[:4], "little") == 42
    assert np.array_equal(np.frombuffer(raw[4:4 + 4 * (n + 1)], dtype=np.int32), rows[0])
    tools.write_samples(f, rows[:3], append=True)
    back = tools.read_samples(f, n)
    assert back.shape == (67, n + 1) and np.array_equal(back[:64], rows) and np.array_equal(back[64:], rows[:3])
    assert np.array_equal(tools.read_samples(f, n, first=10, count=2), rows[10:12])
    f.write_bytes(raw[:1000])
    with pytest.raises(ia.IeacheError):
        tools.read_samples(f, n, count=1)


def test_alice_and_verif_roundtrip(ia, tmp_path):
    """alice.c layout: [sign, bits] under the nbit key, 8 words + zero carry under the secret key."""
    from ieache_amd import tools
    p = ia.default_params().copy(n=10, N=32)
    tools.keygen_files(tmp_path, p)
    value = (1 << 62) | 0xABCDEF  # 64-bit operand (process.c:122-129 sets bit 62)
    tools.alice(tmp_path, sign_code=2, bit_size=64, value=value, seed=5)
    assert (tmp_path / "cloud.data").stat().st_size == 352 * (4 * p.n + 16)  # 11 arrays x 32 (alice.c:167-191)
    tools.alice(tmp_path, sign_code=0, bit_size=32, value=7, seed=6, append=True)
    assert (tmp_path / "cloud.data").stat().st_size == 704 * (4 * p.n + 16)  # what cloud.c:703-766 reads
    # an answer.data has the same 11x32 layout as one operand
    os.rename(tmp_path / "cloud.data", tmp_path / "answer.data")
    code, bits, words = tools.verif(tmp_path)
    assert (code, bits) == (2, 64)
    assert words[:2] == [value & 0xFFFFFFFF, value >> 32] and words[2:] == [0] * 7


def test_verif_interpretation_rules(ia):
    from ieache_amd.tools import verif_interpret as vi
    w = lambda v: [(v >> (32 * i)) & 0xFFFFFFFF for i in range(9)]
    assert vi(1, 0, 32, w(1 << 31)) == 1 << 31            # 2^30 + 2^30 (process.c operands)
    assert vi(1, 4, 32, w(5)) == -5                        # (-A)+(-B)
    assert vi(1, 2, 32, w((3 - 9) & 0xFFFFFFFF)) == -6     # A+(-B) two's complement (verif.c:132-160)
    assert vi(2, 0, 64, w((3 - 9) & (2 ** 64 - 1))) == -6  # A-B
    assert vi(2, 2, 32, w(12)) == 12                       # A-(-B)
    assert vi(2, 1, 32, w(12)) == -12                      # (-A)-B (verif.c:780-783)
    assert vi(4, 0, 64, w(1 << 60)) == 1 << 60             # 2^30 * 2^30
    assert vi(4, 1, 64, w(6)) == -6 and vi(4, 4, 64, w(6)) == 6


def test_c_abi_error_behaviour_without_gpu(ia, tmp_path):
    """Negative errno-style returns + a message, never an exception or a crash (SURVEY 8b conventions)."""
    import ctypes as C
    L = ia.lib()
    # the process contract on a directory with no files: the reference segfaults, we report IEACHE_EIO
    assert L.ieache_cloud_run(os.fsencode(tmp_path)) == -5
    assert b"cloud.key" in L.ieache_last_error() or b"cannot open" in L.ieache_last_error()
    # a context needs a key file ...
    assert not L.ieache_ctx_create(os.fsencode(tmp_path / "nope.key"), 0)
    assert L.ieache_last_error()
    # ... and every entry point tolerates a null context
    assert L.ieache_eval_batch(None, 1, 32, 1, None, None, None) == -22
    assert L.ieache_gates_device(None, 0, 1, None, None, None, None) == -22
    assert L.ieache_ctx_cloud_run(None, b".") == -22
    assert L.ieache_lwe_stride(None) == -22
    L.ieache_ctx_destroy(None)
    # unsupported parameter sets are refused up front
    p = ia.default_params().copy(N=1000)
    assert L.ieache_keygen_raw(C.byref(p), None, 0, None, None, None, None) == -22
    p = ia.default_params().copy(k=2)
    assert L.ieache_keygen_raw(C.byref(p), None, 0, None, None, None, None) == -22
    if ia.device_count() == 0:  # CPU box: there is no fallback path, creating a context must fail loudly
        from ieache_amd import tools
        q = ia.default_params().copy(n=4, N=32)
        k = tools.keygen_raw(q, (1,))
        with pytest.raises(ia.IeacheError):
            ia.Context.from_arrays(q, k["bk"], k["ksk"])


def test_missing_extension_fails_loudly(ia, monkeypatch, tmp_path):
    """The product path must not degrade to anything else when libieache.so is absent."""
    from ieache_amd import evaluator
    monkeypatch.setattr(evaluator, "_LIB", None)
    monkeypatch.setattr(evaluator, "_PKG", str(tmp_path))
    with pytest.raises(ia.IeacheError, match="no CPU fallback"):
        evaluator.lib()
