"""Pin the CPU oracle (oracle/oracle.c) against everything the reference repository offers.

* The README VCF (reference README.md:323-343) is the only golden *output* in the reference repo:
  it pins the MAF column walk of `call` (caller.rs:388-608) incl. cigar_cat_ext_caller/group_by.
* test/test.html (a committed `dotplot` output) holds the base-level segments of record 1 of
  test/testdotplot.paf: it pins the fold over emit_baseplotdatas (cigar.rs:815-952).
* test/test.maf and test/testdotplot.paf are the reference's demo inputs; their expected stat /
  maf2paf / pafcov results were derived by reading the code — those
  assertions document the semantics but are "parity unpinned".
"""
import os

import numpy as np
import pytest

import oracle_py as orc
from helpers import GOLDEN, read_maf_blocks, read_paf


@pytest.fixture(scope="module")
def test_maf():
    blocks = read_maf_blocks(os.path.join(GOLDEN, "test.maf"))
    assert len(blocks) == 1 and len(blocks[0]) == 2
    return blocks[0]


def _call(block, snp, svlen, inv=False, chunk=1000000):
    t, q = block
    return orc.call_var_maf_record(t["name"], q["name"], t["seq"], q["seq"], t["start"], q["start"],
                                   q["align"], q["size"], q["strand"] == "-", snp, inv, svlen, chunk)


def test_readme_vcf_golden(test_maf):
    """`wgatools call test/test.maf -s -l0` — README.md:332-342"""
    golden = [l.rstrip("\n") for l in open(os.path.join(GOLDEN, "readme_call_test_maf_s_l0.vcf"))]
    body = [l for l in golden if not l.startswith("#")]
    got = _call(test_maf, True, 0).splitlines()
    assert len(got) == len(body) == 11
    for g, e in zip(got, body):
        gc, ec = g.split("\t"), e.split("\t")
        assert gc[:8] == ec[:8]  # CHROM POS ID REF ALT QUAL FILTER INFO
        if ec[7] != ".":  # INS / DEL rows: FORMAT + sample are current in the README
            assert gc[8:] == ec[8:]
        else:  # SNP rows: README shows the pre-QI format (stale); current code: caller.rs:582-593
            assert gc[8] == "GT:QI" and gc[9].startswith("1|1:query.chr8@") and gc[9].endswith("@P")


def test_readme_vcf_golden_through_the_paf_caller(test_maf):
    """README.md:317-343 gives ONE output for `call test/test.maf -s -l0` and for `call test/test.paf -s -l0 --target .. --query ..
    -f paf`: the PAF flavour of the caller (caller.rs:610-822) on the CIGAR parse_maf_seq_to_cigar (cigar.rs:344-432) makes of
    the fixture's rows must print the same rows.  That ties two more restatements to a vector the reference holds: the VCF's
    nine SNP, one INS and one DEL positions determine every run length of the CIGAR (109=1D243=1X12=...: eight X ops over nine
    columns, one D, one I), so a wrong CIGAR or a wrong fold would move a row"""
    t, q = test_maf
    golden = [l.rstrip("\n") for l in open(os.path.join(GOLDEN, "readme_call_test_maf_s_l0.vcf")) if not l.startswith("#")]
    _, cg = orc.parse_maf_seq_to_cigar(t["seq"], q["seq"], q["strand"] == "-")
    assert cg == "109=1D243=1X12=1X138=1X177=1X31=1X133=8I18=1X100=2X7=1X22="
    tseq, qseq = t["seq"].replace(b"-", b""), q["seq"].replace(b"-", b"")
    # paf.rs:221-237 fetches [start, end] inclusive: one base behind the aligned stretch
    got = orc.call_within_var_paf(t["name"], q["name"], "cg:Z:" + cg, tseq + b"A", qseq + b"A", t["start"], t["start"] + t["align"],
                                  q["start"], q["start"] + q["align"], q["strand"] == "-", True, 0).splitlines()
    assert len(got) == len(golden) == 11
    for g, e in zip(got, golden):
        gc, ec = g.split("\t"), e.split("\t")
        assert gc[:8] == ec[:8]
        if ec[7] != ".":
            assert gc[8:] == ec[8:]


def test_snp_query_positions(test_maf):
    """Appendix B.5: QI positions of the nine SNP rows"""
    qpos = [int(l.split("\t")[9].split("@")[1]) for l in _call(test_maf, True, 0).splitlines()
            if l.split("\t")[7] == "."]
    assert qpos == [181989773, 181989786, 181989925, 181990103, 181990135, 181990295, 181990396,
                    181990397, 181990405]


def test_chunk_boundaries(test_maf):
    """Appendix B.5 / A.6: cut after the last gap run >= svlen in the window (caller.rs:159-219)"""
    t, q = test_maf[0]["seq"], test_maf[1]["seq"]
    assert orc.find_safe_chunk_boundary(t, q, 0, 1000000, 0) == 857
    assert orc.find_safe_chunk_boundary(t, q, 857, 1000000, 0) == 1008
    assert orc.find_safe_chunk_boundary(t, q, 0, 1000000, 50) == 1008
    cuts, s = [], 0
    while s < 1008:
        e = orc.find_safe_chunk_boundary(t, q, s, 300, 0)
        cuts.append((s, e))
        s = e
    assert cuts == [(0, 110), (110, 410), (410, 710), (710, 857), (857, 1008)]
    # chunked == unchunked event list on this fixture
    assert _call(test_maf, True, 0, chunk=300) == _call(test_maf, True, 0)
    # default flags: no SNPs, indels must be longer than 50
    assert _call(test_maf, False, 50) == ""


def test_stat_maf_counts(test_maf):
    """Appendix B.1/B.2: column CIGAR of test.maf and its counts (cigar.rs:344-432)"""
    counts, txt = orc.parse_maf_seq_to_cigar(test_maf[0]["seq"], test_maf[1]["seq"], False)
    assert txt == "109=1D243=1X12=1X138=1X177=1X31=1X133=8I18=1X100=2X7=1X22="
    assert counts == (990, 9, 1, 8, 1, 1, 0, 0, 0, 0, 0)
    rs = orc.recstat_from(counts)
    assert rs.aligned_size == 1000 and rs.inv_size == 0.0
    # strand '-' routes every indel to the inv_* slots and sets inv_event (cigar.rs:370-403)
    c2, _ = orc.parse_maf_seq_to_cigar(test_maf[0]["seq"], test_maf[1]["seq"], True)
    assert c2 == (990, 9, 0, 0, 0, 0, 1, 8, 1, 1, 1)
    rs2 = orc.recstat_from(c2)
    assert rs2.inv_size == np.float32((1000 + 1007) / 2.0)


def test_stat_paf_fixture():
    """Appendix B.3 (testdotplot.paf): PAF `M` counts as match; '-' strand -> inv_* slots"""
    r1, r2 = read_paf(os.path.join(GOLDEN, "testdotplot.paf"))
    c1 = orc.parse_paf_to_cigar(r1["cg"], r1["strand"] == "-")
    c2 = orc.parse_paf_to_cigar(r2["cg"], r2["strand"] == "-")
    assert c1 == (170, 0, 2, 30, 2, 30, 0, 0, 0, 0, 0)
    assert c2 == (40, 0, 0, 0, 0, 0, 1, 10, 1, 10, 1)
    s2 = orc.recstat_from(c2)
    assert (s2.aligned_size, s2.inv_event, s2.inv_size) == (50, 1, 50.0)


def test_pafcov_fixture():
    """Appendix B.4: only M / = are covered; D and X move without counting (cigar.rs:720-733)"""
    recs = read_paf(os.path.join(GOLDEN, "testdotplot.paf"))
    cov = np.zeros(300, dtype=np.uint64)
    for r in recs:
        orc.update_cov_vec(cov, r["cg"], r["tstart"])
    exp = np.zeros(300, dtype=np.uint64)
    for a, b in ((0, 40), (60, 120), (130, 200), (200, 210), (220, 250)):
        exp[a:b] = 1
    assert (cov == exp).all()
    cov2 = np.zeros(10, dtype=np.uint64)
    orc.update_cov_vec(cov2, "cg:Z:3M2X2D2=5I1S4=", 1)  # clipped at len; X and D skip
    assert cov2.tolist() == [0, 1, 1, 1, 0, 0, 0, 0, 1, 1]


def test_tokeniser_errors():
    """Appendix A.1 — error messages of errors.rs:45-74"""
    def msg(cg):
        with pytest.raises(orc.OracleError) as e:
            orc.parse_paf_to_cigar(cg, False)
        return e.value.message
    assert msg("cg:Z:10M5") == "CIGAR OP `` invalid"            # trailing digits, no op
    assert msg("cg:Z:10MM5I") == "CIGAR OP `MM` invalid"        # op must be one char
    assert msg("cg:Z:M") == "Parse `` Into Integer Error"       # empty length
    assert msg("cg:Z:99999999999999999999M") == "Parse `99999999999999999999` Into Integer Error"
    assert msg("cg:Z:10M3N") == "CIGAR OP `N` invalid"          # stat accepts M = X I D only
    assert msg("cg:Z:10M3é") == "CIGAR OP `é` invalid"  # one multi-byte char is one op
    assert msg("xx:Z:10M3I4M") == "Format error Tag at: xx:Z:10M3I Parse Error by rust::nom, please check"
    assert msg("cg:Z:").startswith("panic")                     # errors.rs:92 slices [..10]
    # first error wins, later ops are not looked at
    assert msg("cg:Z:5Q3") == "CIGAR OP `Q` invalid"


def test_reverse_complement():
    assert orc.reverse_complement(b"ACGTNacgtn") == b"nacgtNACGT"
    with pytest.raises(orc.OracleError) as e:
        orc.reverse_complement(b"ACRGTY")
    assert e.value.message == "Invalid Base: `Y`"  # scanned from the end (utils.rs:85)


def test_insert_semantics():
    """cigar_unit_insert_seq (cigar.rs:492-519): I gaps the target, D gaps the query"""
    t, q = orc.parse_cigar_to_insert("cg:Z:3=2I2X1D2M", b"AAACCGTT", b"AAAGGTTTT")
    assert (t, q) == (b"AAA--CCGTT", b"AAAGGTT-TT")
    # rows are not length-checked: leftover bases stay, a short slice just ends early
    t, q = orc.parse_cigar_to_insert("cg:Z:2M1I", b"ACGT", b"ACG")
    assert (t, q) == (b"AC-GT", b"ACG")
    t, q = orc.parse_cigar_to_insert("cg:Z:10M", b"ACGT", b"ACG")
    assert (t, q) == (b"ACGT", b"ACG")
    with pytest.raises(orc.OracleError) as e:  # insertion point beyond the string: panic
        orc.parse_cigar_to_insert("cg:Z:6M1I", b"ACGT", b"ACGTACG")
    assert e.value.kind == 6


def test_pseudo_maf_by_cigar():
    """gen_pesudo_maf_by_cigar (cigar.rs:744-804)"""
    assert orc.gen_pesudo_maf_by_cigar("cg:Z:3=2I2X1D2M1S", b"AAAGGTTCCx", True) == b"AAATT-CC"
    assert orc.gen_pesudo_maf_by_cigar("cg:Z:3=2I2X1D2M4N", b"", False) == b"11100-11"


def test_cs_to_cigar():
    """paf.rs:154-158 doc example"""
    assert orc.cs_to_cigar(":6-ata:10+gtc:4*at*tg:3") == "6M3D10M3I4M2X3M"
    assert orc.cs_to_cigar(":5=ACGT*ag:2") == "5M1X2M"


def test_dotplot_test_html_golden():
    """the data rows of the reference's committed test/test.html = `dotplot` base-level segments of record 1 of
    test/testdotplot.paf (the file predates the second record and the cutoff default of 50: any cutoff below the
    record's shortest indel, 10, reproduces it)"""
    import json
    want = json.load(open(os.path.join(GOLDEN, "test_html_values.json")))
    line = open(os.path.join(GOLDEN, "testdotplot.paf")).read().splitlines()[0].split("\t")
    cg = [t for t in line[12:] if t.startswith("cg:Z:")][0]
    for cutoff in (0, 9):
        segs = orc.cigar_to_base_plotdata(cg, int(line[7]), int(line[2]), line[4] == "-", cutoff)
        got = [dict(cigar="MID"[int(s[4])], query_chro=line[0], query_end=int(s[3]), query_start=int(s[2]),
                    ref_chro=line[5], ref_end=int(s[1]), ref_start=int(s[0])) for s in segs]
        assert got == want
