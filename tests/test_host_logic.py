"""Host-side logic and the C-ABI surface -- CPU only (no kernel is launched here)."""
import ctypes
import os
import re
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def norm(sets):
    return sorted([[s, e, sorted(k)] for s, e, k in sets], key=lambda x: (x[0], -1 if x[1] is None else x[1]))


def test_kmer_heuristic_matches_reference_golden(golden):
    from cutadapt_amd.kmer_heuristic import create_positions_and_kmers
    cases = golden("heuristic.json")
    assert len(cases) >= 300
    for c in cases:
        got = create_positions_and_kmers(c["adapter"], c["min_overlap"], c["error_rate"],
                                         c["back"], c["front"], c["internal"])
        assert norm(got) == c["result"], c


def test_kmer_heuristic_known_sets():
    """the TruSeq / e=0.1 / O=3 search sets quoted in SURVEY.md section 8(a) a10"""
    from cutadapt_amd.kmer_heuristic import create_positions_and_kmers, kmer_chunks
    assert kmer_chunks("AABCABCABC", 3) == {"AABC", "ABC"}         # reference kmer_heuristic.py:8-9
    sets = create_positions_and_kmers("AGATCGGAAGAGCACACGTCTGAACTCCAGTCA", 3, 0.1, True, False)
    d = {(s, e): sorted(k) for s, e, k in sets}
    assert d[(-3, None)] == ["AGA"] and d[(-4, None)] == ["AGAT"]
    assert d[(-19, None)] == ["AGATC", "GGAAG"]
    assert len(d[(-29, None)]) == 3 and len(d[(-33, None)]) == 4 and len(d[(0, None)]) == 4
    assert "".join(sorted(d[(0, None)], key="AGATCGGAAGAGCACACGTCTGAACTCCAGTCA".index)) == \
        "AGATCGGAAGAGCACACGTCTGAACTCCAGTCA"
    # front adapters mirror the windows to the read start
    fs = create_positions_and_kmers("ACGTACGTAC", 3, 0.1, False, True)
    assert all(s == 0 for s, _, _ in fs)


def test_library_is_built_and_exports_every_declared_symbol():
    """every function include/cutadapt_hip.h declares is exported by the in-tree .so"""
    from cutadapt_amd import _lib
    header = open(os.path.join(ROOT, "include", "cutadapt_hip.h")).read()
    declared = set(re.findall(r"\b(cah_[a-z0-9_]+)\s*\(", header))
    declared -= {"cah_plan", "cah_kmer_set", "cah_adapter_desc"}
    assert declared == set(_lib.EXPORTED_SYMBOLS), declared ^ set(_lib.EXPORTED_SYMBOLS)
    assert os.path.exists(_lib.LIB_PATH), "build the library first: python -m cutadapt_amd.build"
    L = _lib.lib()
    for name in declared:
        assert hasattr(L, name), name
    assert L.cah_abi_version() == _lib.ABI_VERSION == 5
    # the binary says which sources it was built from (cah_build_id), and build.needs_build() goes by that, not by file times
    from cutadapt_amd import build
    assert _lib.build_id() == build.source_hash() == build.library_build_id(), "stale library: python -m cutadapt_amd.build"
    assert not build.needs_build()
    out = subprocess.run(["nm", "-D", "--defined-only", _lib.LIB_PATH], capture_output=True, text=True).stdout
    exported = set(re.findall(r" T (cah_[a-z0-9_]+)", out))
    assert declared <= exported
    # the hot-path library must not depend on the oracle or on libtorch
    ldd = subprocess.run(["ldd", _lib.LIB_PATH], capture_output=True, text=True).stdout
    assert "oracle" not in ldd and "torch" not in ldd


def test_product_never_imports_the_oracle():
    for dirpath, _, files in os.walk(os.path.join(ROOT, "cutadapt_amd")):
        for f in files:
            if f.endswith((".py", ".cpp", ".hip", ".h")):
                src = open(os.path.join(dirpath, f), errors="replace").read()
                assert not re.search(r"^\s*(from|import)\s+oracle", src, re.M), f
                assert "cutadapt_oracle" not in src and "orc_" not in src, f


def test_plan_validation_matches_reference_errors():
    """constructor errors are raised on the host, before any device is touched
    (reference _align.pyx:217-218, :270-271, :631-636; _kmer_finder.pyx:133-140)"""
    from cutadapt_amd import _lib
    S = _lib.MatcherSpec
    with pytest.raises(ValueError, match="only N wildcards"):
        _lib.Plan([S("NNNN", 0.1, wildcard_ref=True)])
    with pytest.raises(ValueError, match="indel_cost"):
        _lib.Plan([S("ACGT", 0.1, indel_cost=0)])
    with pytest.raises(ValueError, match="between 0 and 1"):
        _lib.Plan([S("ACGT", 1.5, kind=_lib.KIND_PREFIX)])
    with pytest.raises(ValueError, match="min_overlap"):
        _lib.Plan([S("ACGT", 0.1, min_overlap=0, kind=_lib.KIND_SUFFIX)])
    with pytest.raises(ValueError, match="longer than the maximum"):
        _lib.Plan([S(kind=_lib.KIND_KMER_ONLY, kmer_sets=[(0, None, ["A" * 65])])])
    with pytest.raises(TypeError):
        _lib.Plan([S(kind=_lib.KIND_KMER_ONLY, kmer_sets=[(0, None, [b"ACGT"])])])
    with pytest.raises(ValueError, match="ASCII"):
        _lib.Plan([S("ACGÜ", 0.1)])
    # adapters longer than 64 characters are accepted (reference _align.pyx:250-257: any length) and ask
    # for column scratch in the plan's workspace
    long_plan = _lib.Plan([S("ACGT" * 40, 0.1)])
    L = _lib.lib()
    assert L.cah_plan_workspace_bytes(long_plan.handle, 1000) > L.cah_workspace_bytes(1000)
    with pytest.raises(_lib.UnsupportedByHipPath):
        _lib.Plan([S("A" * 100001, 0.1)])
    # valid plans are built on the host without a GPU; tables are uploaded lazily per device
    plan = _lib.Plan([S("AGGNNNNNNNNNNNNNNTTC", 0.1, 14, wildcard_ref=True, min_overlap=3),
                      S("CNNNNNNNNGTT", 0.25, wildcard_ref=True, kind=_lib.KIND_PREFIX),
                      S(kind=_lib.KIND_KMER_ONLY,
                        kmer_sets=[(0, None, ["ACGT" * 10] * 3), (-5, None, ["AC", "GT"])])])
    assert plan.effective_length(0) == 6          # reference tests/test_align.py:336
    assert plan.effective_length(1) == 4
    assert plan.n_kmer_entries(2) == 4            # 3 x 40 chars need 3 words (+1 for the second set)
    assert plan.n_kmer_entries(0) == 0


def test_no_silent_cpu_fallback_without_gpu():
    """without a HIP device the batch entry points fail loudly"""
    from cutadapt_amd import _lib
    if _lib.device_count() > 0:
        pytest.skip("a GPU is present")
    from cutadapt_amd.align import Aligner
    a = Aligner("ACGTACGT", 0.1)               # host-only construction works
    with pytest.raises(RuntimeError):
        a.locate("TTACGTACGTTT")


def test_pack_strings_and_ascii_check():
    from cutadapt_amd.batch import pack_strings
    seqs, offsets = pack_strings(["ACGT", "", "TTGCA"])
    assert offsets.tolist() == [0, 4, 4, 9] and bytes(seqs) == b"ACGTTTGCA"
    with pytest.raises(ValueError):
        pack_strings(["ACGÜ"])
    with pytest.raises(TypeError):
        pack_strings([b"ACGT"])


def test_adapter_parameter_normalisation():
    """SingleAdapter.__init__ rules (reference adapters.py:577-595) -- pure host logic"""
    from cutadapt_amd import adapters as A
    a = A.BackAdapter("acgun", max_errors=2, min_overlap=10)
    assert a.sequence == "ACGTN" and a.min_overlap == 5
    assert a.max_error_rate == 2 / 4           # absolute errors / non-N characters
    assert a.adapter_wildcards is True
    assert A.BackAdapter("ACGT").adapter_wildcards is False      # plain ACGT: no wildcard mode
    with pytest.raises(ValueError):
        A.BackAdapter("")
    with pytest.raises(A.InvalidCharacter):
        A.BackAdapter("ACGZ")
    assert A.Where.BACK == 14 and A.Where.FRONT == 11 and A.Where.PREFIX == 8 and A.Where.SUFFIX == 2
    assert A.Where.FRONT_NOT_INTERNAL == 9 and A.Where.BACK_NOT_INTERNAL == 6 and A.Where.ANYWHERE == 15
    p = A.PrefixAdapter("ACGTACGT", indels=False)
    assert type(p.aligner).__name__ == "PrefixComparer" and isinstance(p.kmer_finder, A.MockKmerFinder)
    assert type(A.SuffixAdapter("ACGTACGT").aligner).__name__ == "Aligner"
    long_ad = A.BackAdapter("ACGT" * 16)            # 64 chars: k-mers fit
    assert long_ad.kmer_finder is not None
    m = A.RemoveAfterMatch(0, 4, 10, 14, 4, 0, adapter=a, sequence="ACGTACGTAAACGTTT")
    assert m.trimmed("ACGTACGTAAACGTTT") == "ACGTACGTAA" and m.rest() == "TT"
    assert m.remainder_interval() == (0, 10) and m.removed_sequence_length() == 6


def test_empty_kmers_are_accepted_like_the_reference():
    """an adapter that may have as many errors as it has characters: kmer_heuristic emits an empty k-mer; the reference's
    KmerFinder takes it (it is never found, _kmer_finder.pyx:121-160) -- so does this one, and the library searches the sets
    without it"""
    from cutadapt_amd import adapters as A
    from cutadapt_amd._kmer_finder import KmerFinder
    sets = [(0, None, ["", "A"]), (-3, None, [""])]
    f = KmerFinder(sets)
    assert f.positions_and_kmers == sets and f.searched_positions_and_kmers == [(0, None, ["A"]), (-3, None, [])]
    assert f.number_of_searches == 1
    ad = A.BackAdapter("ACGT", max_errors=4, min_overlap=3)
    assert ad.max_error_rate == 1.0 and isinstance(ad.kmer_finder, KmerFinder)
    assert any("" in k for _, _, k in ad.kmer_finder.positions_and_kmers)
    assert all("" not in k for _, _, k in ad.matcher_spec().kmer_sets)


def test_prefilter_kernel_selection():
    """which prefilter kernel a plan gets (host-side decision of cah_plan_create): everything
    kmer_heuristic builds is 'lean', hand-made windows and long k-mers stay 'general'"""
    from cutadapt_amd import adapters as A
    from cutadapt_amd._kmer_finder import KmerFinder
    T = "AGATCGGAAGAGCACACGTCTGAACTCCAGTCA"
    assert A.BackAdapter(T)._fused_plan.prefilter_kind() == "lean"
    assert A.FrontAdapter(T)._fused_plan.prefilter_kind() == "lean"
    assert A.AnywhereAdapter(T)._fused_plan.prefilter_kind() == "lean"
    assert A.NonInternalBackAdapter(T)._fused_plan.prefilter_kind() == "lean"
    assert A.BackAdapter("ACGTNNACGTRYACGT", max_errors=0.2)._fused_plan.prefilter_kind() == "lean"      # IUPAC adapters too
    assert A.PrefixAdapter(T, indels=False)._fused_plan.prefilter_kind() == "none"                      # comparer: MockKmerFinder
    assert KmerFinder([(5, 40, ["ACGTACG"])])._plan.prefilter_kind() == "lean"                          # head window inside the span
    assert KmerFinder([(5, 80, ["ACGTACG"])])._plan.prefilter_kind() == "general"                       # window beyond 64 characters
    assert KmerFinder([(0, -3, ["ACGTACG"])])._plan.prefilter_kind() == "general"                       # negative stop
    assert KmerFinder([(0, None, ["A" * 40])])._plan.prefilter_kind() == "general"                      # k-mer longer than 32
    assert KmerFinder([(-80, None, ["ACGT"])])._plan.prefilter_kind() == "general"                      # tail window beyond the span


def test_adapter_specifications_follow_the_reference_grammar():
    """per-adapter search parameters, placement restrictions, brace repeats and linked adapters
    (reference parser.py:28-86, :203-300, :472-551; known answers of reference tests/test_parser.py)"""
    from cutadapt_amd import adapters as A
    from cutadapt_amd.pipeline import adapter_from_spec, expand_braces, parse_search_parameters
    assert parse_search_parameters("e=0.1") == {"max_errors": 0.1}
    assert parse_search_parameters("error_rate=0.1") == {"max_errors": 0.1}
    assert parse_search_parameters("max_errors=2") == {"max_errors": 2}
    assert parse_search_parameters("o=5") == {"min_overlap": 5}
    assert parse_search_parameters("o=7; e=0.4") == {"min_overlap": 7, "max_errors": 0.4}
    assert parse_search_parameters("anywhere") == {"anywhere": True}
    assert parse_search_parameters("required") == {"required": True}
    assert parse_search_parameters("optional") == {"required": False}
    assert parse_search_parameters("noindels") == {"indels": False}
    assert parse_search_parameters("indels") == {"indels": True}
    assert parse_search_parameters("rightmost") == {"rightmost": True}
    for bad, exc in (("e=hallo", ValueError), ("bla=0.1", KeyError), ("e=", ValueError), ("e=0.1;e=0.1", KeyError),
                     ("e=0.1;max_errors=0.1", KeyError), ("optional; required", ValueError), ("indels; noindels", ValueError)):
        with pytest.raises(exc):
            parse_search_parameters(bad)
    assert expand_braces("TGA{5}CT") == "TGAAAAACT" and expand_braces("GG{2}TGA{5}CT") == "GGGTGAAAAACT"
    assert expand_braces("A{0}") == "" and expand_braces("TGA{0}CT") == "TGCT"
    for bad in ("{", "}", "{}", "{5", "{1}", "A{-7}", "A{", "A{1", "N{7", "AN{7", "A{4{}", "A{4}{3}", "A{b}", "A{6X}", "A{X6}", "A}A"):
        with pytest.raises(ValueError):
            expand_braces(bad)
    d = dict(max_errors=0.1, min_overlap=3)
    a = adapter_from_spec("a_name=ADAPTER;e=0.2;o=5".replace("ADAPTER", "ACGTACGTAC"), "back", **d)
    assert type(a) is A.BackAdapter and a.name == "a_name" and a.max_error_rate == 0.2 and a.min_overlap == 5
    assert type(adapter_from_spec("ACGT;noindels", "back", **d)) is A.BackAdapter and not adapter_from_spec("ACGT;noindels", "back", **d).indels
    assert adapter_from_spec("ACGTACGT;anywhere", "back", **d)._force_anywhere
    assert type(adapter_from_spec("ACGTACGT;rightmost", "front", **d)) is A.RightmostFrontAdapter
    assert type(adapter_from_spec("ACGTACGT;rightmost", "back", **d)) is A.RightmostBackAdapter
    assert type(adapter_from_spec("^ACGT", "front", **d)) is A.PrefixAdapter
    assert type(adapter_from_spec("ACGT$", "back", **d)) is A.SuffixAdapter
    assert type(adapter_from_spec("XACGT", "front", **d)) is A.NonInternalFrontAdapter
    assert type(adapter_from_spec("ACGTX", "back", **d)) is A.NonInternalBackAdapter
    assert type(adapter_from_spec("ACGT...", "back", **d)) is A.FrontAdapter
    assert type(adapter_from_spec("...ACGT", "back", **d)) is A.BackAdapter
    assert adapter_from_spec("A{10}C", "back", **d).sequence == "AAAAAAAAAAC"
    assert adapter_from_spec("ACGT;o=15", "back", **d).min_overlap == 4            # clamped to the adapter length
    for bad, t in (("^ACGT;o=3", "front"), ("^ACGT$", "front"), ("^XACGT", "front"), ("ACGT$", "front"), ("^ACGT", "back"),
                   ("^ACGT", "anywhere"), ("ACGT;rightmost", "anywhere"), ("^ACGT;rightmost", "front"),
                   ("ACGT;required", "back"), ("ACGT...TGCA", "anywhere"), ("...ACGT", "front")):
        with pytest.raises(ValueError):
            adapter_from_spec(bad, t, **d)
    linked = adapter_from_spec("ACGT;o=2...TTTTAAAA;e=0.3", "back", **d)
    assert type(linked) is A.LinkedAdapter and not linked.front_required and not linked.back_required
    assert linked.front_adapter.min_overlap == 2 and linked.back_adapter.max_error_rate == 0.3
    linked = adapter_from_spec("^ACGT...TTTTAAAA;required", "back", **d)
    assert linked.front_required and linked.back_required and type(linked.front_adapter) is A.PrefixAdapter
    linked = adapter_from_spec("ACGT;optional...TTTTAAAA", "front", **d)
    assert not linked.front_required and linked.back_required
    linked = adapter_from_spec("name=ACGT...TTTTAAAA$", "back", **d)
    assert linked.name == "name" and not linked.front_required and linked.back_required


def test_stream_copy_plan_division_is_exact():
    """k_filter_stream (csrc/kernels.hip) maps the 16-byte unit u of a wave's piece to read u // U and unit u % U
    with a multiply: r = (u * ceil(65536 / U)) >> 16.  Exhaustive over the kernel's range (u < 64 * 16, U <= 16)."""
    for U in range(1, 17):
        magic = (65536 + U - 1) // U
        for u in range(64 * 16):
            r = (u * magic) >> 16
            assert r == u // U and u * magic < 2 ** 32, (U, u)


def test_end_aligned_copy_plan_of_views():
    """k_filter_stream2's RV form (csrc/stream2.hip): the copy plan that streams views of a uniform batch END-aligned,
    replayed in numpy.  Per read r of a piece, lane r packs A = r (n - 16 H) + S2_RV_BACK - d | T << 16 with
    T = skip + 16 r H for either half's shape; unit u = 64 k + lane of a half takes its read's word, is skipped (zeros) iff
    its last character 16 u + 15 [+ 16 H1] < T, else fetched from resource byte A + 16 lane + 1024 k [+ 16 H1] (resource
    base = the piece's first byte - S2_RV_BACK) into slot row r, unit u % H.  Checked for every read length of the short
    form: both fields fit 16 bits, no offset is negative, and what lands in the slot -- with the unit a view starts in
    masked and everything from position n on cleared, as the kernel does -- is the view, NUL-padded in front, ending at n."""
    rng = np.random.default_rng(11)
    BACK, WAVE = 1184, 64
    for n in list(range(1, 161)):
        U = (n + 15) // 16
        H1, H2 = (U + 1) // 2, U - (U + 1) // 2
        piece = rng.integers(65, 91, size=WAVE * n, dtype=np.uint8)
        before = rng.integers(97, 123, size=BACK, dtype=np.uint8)            # bytes in front of the piece (other reads)
        after = rng.integers(97, 123, size=64, dtype=np.uint8)
        mem = np.concatenate([before, piece, after])                         # resource byte i = mem[i]
        start = rng.integers(0, n + 1, size=WAVE)
        length = np.array([rng.integers(0, n - s + 1) for s in start])
        if n > 3:
            start[:4], length[:4] = [0, 0, n, 1], [n, 0, 0, n - 1]
        d, skip = n - (start + length), n - length
        rows = np.zeros((WAVE, 160), dtype=np.uint8)
        for half, H, base in ((0, H1, 0), (1, H2, 16 * H1)):
            if H == 0:
                continue
            r_lane = np.arange(WAVE)
            A = r_lane * (n - 16 * H) + BACK - d
            T = skip + 16 * r_lane * H
            assert (A >= 0).all() and (A < 65536).all() and (T >= 0).all() and (T < 65536).all(), (n, half)
            magic = (65536 + H - 1) // H
            for k in range(H):
                for lane in range(WAVE):
                    u = k * WAVE + lane
                    r = (u * magic) >> 16
                    assert r == u // H and r < WAVE
                    c = u - r * H
                    front = 16 * lane + 16 * k * WAVE + 15 + base < T[r]
                    assert front == (16 * c + base + 15 < skip[r])
                    if front:
                        unit = np.zeros(16, dtype=np.uint8)
                    else:
                        off = A[r] + 16 * lane + 1024 * k + base
                        assert 0 <= off and off + 16 <= len(mem), (n, half, k, lane, off)
                        unit = mem[off:off + 16]
                    rows[r, base + 16 * c: base + 16 * c + 16] = unit
        for r in range(WAVE):
            got = rows[r].copy()
            got[:skip[r]] = np.where(np.arange(skip[r]) // 16 == skip[r] // 16, 0, got[:skip[r]])   # the unit the view starts in
            assert not got[:skip[r]].any(), (n, r)                           # ... everything in front of it arrived as zeros
            got[n:] = 0                                                      # finish(): characters past the read's end
            want = np.zeros(160, dtype=np.uint8)
            want[skip[r]:n] = piece[r * n + start[r]: r * n + start[r] + length[r]]
            assert np.array_equal(got, want), (n, r, start[r], length[r])


def test_prefilter_of_anchored_exact_adapters_is_dropped_only_when_implied():
    """CahMatcher::filter_implied (cah_plan_create): the prefilter of an anchored adapter that tolerates no error is
    skipped only if an unedited occurrence of the adapter at its anchored place passes it -- decided against the
    plan's own k-mer sets, windows and character tables, not assumed from how cutadapt builds them."""
    import ctypes as C
    import struct
    from cutadapt_amd import _lib
    from cutadapt_amd import adapters as A

    def implied(plan):
        L = _lib.lib()
        need = C.c_size_t(0)
        _lib.check(L.cah_plan_debug_matcher(plan.handle, 0, None, 0, C.byref(need)))
        buf = (C.c_uint8 * need.value)()
        _lib.check(L.cah_plan_debug_matcher(plan.handle, 0, buf, need.value, C.byref(need)))
        return struct.unpack_from("<i", bytes(buf), need.value - 8)[0]

    PREFIX, SUFFIX, BACK = 8, 2, 14
    # what the adapter classes build
    assert implied(A.PrefixAdapter("NNNNNNNNACGTACGT", max_errors=0.1)._fused_plan) == 1
    assert implied(A.SuffixAdapter("ACGTACGTAC", max_errors=0.05)._fused_plan) == 1
    assert implied(A.PrefixAdapter("ACGTACGTACGTACGTACGT", max_errors=0.1)._fused_plan) == 1    # (implied, but k = 2: the DP stays)
    assert implied(A.BackAdapter("ACGTACGTACGT")._fused_plan) == 0                               # not anchored

    def plan(seq, flags, sets, wr=False, kwr=False):
        return _lib.Plan([_lib.MatcherSpec(seq, 0.0, flags, wr, False, 1, len(seq), kmer_sets=sets, kmer_ref_wildcards=kwr)])

    assert implied(plan("ACGTACGT", PREFIX, [(0, None, ["TTTT"])])) == 0          # not a piece of the adapter
    assert implied(plan("ACGTACGT", PREFIX, [(0, None, ["GTAC"])])) == 1
    assert implied(plan("ACGTACGT", PREFIX, [(4, 12, ["ACGT"])])) == 1           # the copy at 4 lies inside [4, 12)
    assert implied(plan("ACGTACGT", PREFIX, [(5, 12, ["ACGT"])])) == 0           # no copy starts at >= 5
    assert implied(plan("ACGTACGT", PREFIX, [(0, 7, ["TACGT"])])) == 0           # the copy ends at 8 > stop
    assert implied(plan("ACGTACGT", SUFFIX, [(-4, None, ["ACGT"])])) == 1        # last four characters
    assert implied(plan("ACGTACGT", SUFFIX, [(-3, None, ["ACGT"])])) == 0
    assert implied(plan("ACGTACGT", SUFFIX, [(0, 8, ["ACGT"])])) == 0            # a head window says nothing about the read end
    assert implied(plan("ACGTACGT", BACK, [(0, None, ["ACGT"])])) == 0
    # the aligner accepts ANY byte at an N of the adapter (NUL included), a k-mer table never matches NUL
    # (_match_tables.py:73-78) and, without adapter wildcards, only 'N' itself: a k-mer across the N proves nothing
    assert implied(plan("ACGNACGT", PREFIX, [(0, None, ["CGNA"])], wr=True, kwr=False)) == 0
    assert implied(plan("ACGNACGT", PREFIX, [(0, None, ["CGNA"])], wr=True, kwr=True)) == 0
    assert implied(plan("ACGNACGT", PREFIX, [(0, None, ["ACGT"])], wr=True, kwr=True)) == 1      # the piece behind the N does


def test_which_way_through_the_device_fastq_path():
    """host only: the option sets the all-device way of gpu_pipeline serves (everything else goes the general way)"""
    from cutadapt_amd import adapters as A
    from cutadapt_amd.gpu_pipeline import _all_device_adapters, _index_regroups, _mate_all_device
    back, front = A.BackAdapter("AGATCGGAAGAGC"), A.FrontAdapter("TGGAATTCTCGG")
    p1, p2, s1 = A.PrefixAdapter("ACGTACGT"), A.PrefixAdapter("TTGCAATG"), A.SuffixAdapter("GGCCAATT")
    linked = A.LinkedAdapter(A.PrefixAdapter("ACGTACGT"), A.BackAdapter("AGATCGGAAGAGC"), True, False, "l")
    rightmost = A.RightmostFrontAdapter("TGGAATTCTCGG")
    linked_rm = A.LinkedAdapter(A.RightmostFrontAdapter("TGGAATTCTCGG"), A.BackAdapter("AGATCGGAAGAGC"), True, False, "lr")
    assert _all_device_adapters([back], 1, True) and _all_device_adapters([back, front, p1, s1], 3, True)
    assert not _all_device_adapters([back, rightmost], 1, True)
    # more than one anchored 5' (or 3') adapter: AdapterCutter(index=True) regroups them behind an AdapterIndex
    assert _index_regroups([p1, p2]) and not _index_regroups([p1, s1]) and not _index_regroups([back, front])
    assert not _all_device_adapters([p1, back, p2], 1, True) and _all_device_adapters([p1, back, p2], 1, False)
    # one linked adapter of non-rightmost parts, one round
    assert _all_device_adapters([linked], 1, True)
    assert not _all_device_adapters([linked], 2, True) and not _all_device_adapters([linked, back], 1, True)
    assert not _all_device_adapters([linked_rm], 1, True)
    # a mate's BatchTrimmer options -> (adapters, pre, post, times) or None
    got = _mate_all_device(dict(adapters=[back, front], times=2, quality_cutoff=(0, 20), cut=[3, 0, -2], poly_a=True, length=50))
    assert got is not None and got[3] == 2 and got[1]["cut"] == [3, -2] and got[2]["poly_a"] and got[2]["length"] == 50
    assert _mate_all_device(dict(adapters=[back])) == ([back], None, None, 1)
    assert _mate_all_device(dict()) == ([], None, None, 1)                    # a mate without adapters or modifiers
    assert _mate_all_device(dict(adapters=[linked], nextseq_trim=20))[1]["nextseq_trim"] == 20
    for opts in (dict(adapters=[back], action="mask"), dict(adapters=[back], revcomp=True), dict(adapters=[linked], times=2),
                 dict(adapters=[p1, p2]), dict(adapters=[back], cut=[1, 2]), dict(adapters=[back], cut=[1, -2, 3]),
                 dict(adapters=[back], some_new_option=1)):
        assert _mate_all_device(opts) is None, opts
    assert _mate_all_device(dict(adapters=[p1, p2], index=False)) is not None
