"""CPU: checks on the BUILT gfx950 code of libdau_conv_hip.so (no GPU needed: the code objects are taken out of the shared
library's .hip_fatbin section and disassembled with llvm-objdump).

Round 3 lost two GPU runs to hazards that only exist in the machine code: (a) an SGPR base reloaded by v_readlane_b32 right in
front of an inline-asm global_load (a VALU write of an SGPR needs 5 wait states before a vector memory instruction reads it; hipcc's
hazard recognizer does not look into asm blocks) and (b) inline-asm loads whose destination registers the allocator handed to
something else before the hand-counted s_waitcnt.  Nothing in the sources shows either; this file looks at what ships:
  1. no VALU-written SGPR is read by a vector memory instruction within 5 wait states,
  2. no register with a vector-memory load in flight is read or written before an s_waitcnt vmcnt that covers that load
     (straight-line code only -- the state is dropped at every unconditional branch, so a hazard across a join is not seen;
     the round-3 failures were both inside unrolled straight-line blocks; loads return in order, stores only make a counted
     wait stricter),
  3. the hot kernels touch no scratch between their first and their last MFMA (spills in prologue / flush code are tolerated,
     they are listed), and the kernels that are not instantiated for production are not in the release library at all,
  4. every kernel of the release library is byte-identical in the tuning build (the variant tests run on that build)."""
import os
import re
import struct
import subprocess
import tempfile

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "dau-convnet_amd", "dau_conv")
LLVM = "/opt/rocm/llvm/bin"
HOT = ("gather_dot_kernel", "gather_mfma_kernel", "wg_gemm_kernel", "dense_gather_kernel", "split_gather_kernel")


def _code_objects(so, outdir):
    """the gfx950 code objects of every translation unit of `so` -> list of file paths"""
    fat = os.path.join(outdir, os.path.basename(so) + ".fatbin")
    subprocess.check_call([os.path.join(LLVM, "llvm-objcopy"), "--dump-section", ".hip_fatbin=" + fat, so, os.path.join(outdir, "discard.so")])
    data = open(fat, "rb").read()
    magic = b"__CLANG_OFFLOAD_BUNDLE__"
    out, off = [], data.find(magic)
    while off >= 0:
        n = struct.unpack_from("<Q", data, off + 24)[0]
        p = off + 32
        for _ in range(n):
            o, s, ts = struct.unpack_from("<QQQ", data, p)
            p += 24
            triple = data[p:p + ts].decode()
            p += ts
            if "gfx950" in triple and s:
                path = os.path.join(outdir, "%s.%d.co" % (os.path.basename(so), len(out)))
                open(path, "wb").write(data[off + o:off + o + s])
                out.append(path)
        off = data.find(magic, off + 1)
    return out


def _functions(co):
    """{symbol: [(mnemonic, operand string)]} of one code object"""
    txt = subprocess.run([os.path.join(LLVM, "llvm-objdump"), "-d", "--no-show-raw-insn", co], capture_output=True, text=True, check=True).stdout
    funcs, cur = {}, None
    for line in txt.splitlines():
        m = re.match(r"^[0-9a-f]+ <(.+)>:$", line)
        if m:
            cur = funcs.setdefault(m.group(1), [])
            continue
        if cur is None or not line.startswith("\t"):
            continue
        body = line.split("//")[0].strip()
        if not body:
            continue
        parts = body.split(None, 1)
        cur.append((parts[0], parts[1] if len(parts) > 1 else ""))
    return funcs


def _regs(text, kind):
    """register numbers of class `kind` ('s', 'v', 'a') named in an operand string"""
    out = set()
    for m in re.finditer(r"(?<![a-z0-9_])%s(\d+)(?![\d:])|(?<![a-z0-9_])%s\[(\d+):(\d+)\]" % (kind, kind), text):
        if m.group(1) is not None:
            out.add(int(m.group(1)))
        else:
            out.update(range(int(m.group(2)), int(m.group(3)) + 1))
    return out


def _is_vmem(mn):
    return mn.startswith(("global_", "buffer_", "flat_", "scratch_"))


def _kernel_name(sym):
    out = subprocess.run(["c++filt", sym], capture_output=True, text=True).stdout.strip()
    return out or sym


@pytest.fixture(scope="module")
def release():
    so = os.path.join(PKG, "libdau_conv_hip.so")
    if not os.path.exists(so):
        pytest.skip("library not built")
    with tempfile.TemporaryDirectory() as d:
        funcs = {}
        for co in _code_objects(so, d):
            funcs.update(_functions(co))
        yield funcs


def _sgpr_hazards(ins):
    """(instruction, sgprs, age) for every vector memory instruction that reads an SGPR a VALU instruction wrote < 5 wait states ago"""
    bad, recent = [], []          # recent: (wait states since, sgprs written) of VALU instructions that wrote SGPRs
    for mn, ops in ins:
        if _is_vmem(mn):
            used = _regs(ops, "s")
            for age, regs in recent:
                if age < 5 and used & regs:
                    bad.append(("%s %s" % (mn, ops), sorted(used & regs), age))
        step = int(ops.strip() or "0", 0) + 1 if mn == "s_nop" else 1
        recent = [(a + step, r) for a, r in recent if a + step < 5]
        if mn.startswith("v_"):
            w = _regs(ops.split(",")[0], "s")
            if w:
                recent.append((0, w))
    return bad


def _inflight_hazards(ins):
    """(instruction, registers, load) for every instruction that touches the destination of a vector-memory load no s_waitcnt has
    covered yet.  Straight-line code only: the state is dropped where the next instruction is not reached by falling through."""
    bad, inflight = [], []        # inflight: loads in issue order: (dest VGPR/AGPR set, text)
    for mn, ops in ins:
        if mn in ("s_endpgm", "s_branch", "s_setpc_b64"):
            inflight = []
            continue
        if mn == "s_waitcnt":
            m = re.search(r"vmcnt\((\d+)\)", ops)
            if m:
                n = int(m.group(1))
                # loads return in order: with at most n operations outstanding, all but the newest n LOADS are complete
                # (stores issued in between only make the wait stricter)
                inflight = inflight[len(inflight) - n:] if n else []
            continue
        touched = {("v", r) for r in _regs(ops, "v")} | {("a", r) for r in _regs(ops, "a")}
        for dest, text in inflight:
            if touched & dest:
                bad.append(("%s %s" % (mn, ops), sorted(touched & dest)[:4], text))
        if _is_vmem(mn) and "load" in mn:
            if "_lds_" in mn or mn.endswith("_lds"):
                inflight.append((set(), mn))          # LDS DMA: no register, but a place in the in-order queue of loads
                continue
            first = ops.split(",")[0]
            dest = {("v", r) for r in _regs(first, "v")} | {("a", r) for r in _regs(first, "a")}
            if dest:
                inflight.append((dest, "%s %s" % (mn, ops)))
    return bad


def test_the_checkers_see_the_two_round_three_hazards():
    """(a) k_gather_dot.hip before its s_mov_b64 fix; (b) k_dense_wgrad.hip's look-ahead load past the end of a row"""
    a = [("v_readlane_b32", "s6, v127, 3"), ("v_readlane_b32", "s7, v127, 4"), ("global_load_dwordx2", "v[56:57], v92, s[6:7]")]
    assert _sgpr_hazards(a) and not _sgpr_hazards(a[:2] + [("s_nop", "4")] + a[2:])
    assert not _sgpr_hazards(a[:2] + [("s_mov_b64", "s[8:9], s[6:7]"), ("global_load_dwordx2", "v[56:57], v92, s[8:9]")])
    b = [("global_load_dwordx4", "v[162:165], v1, s[38:39]"), ("global_load_lds_dwordx4", "v[2:3], off"),
         ("ds_read_b128", "v[162:165], v169 offset:10240")]
    assert _inflight_hazards(b)
    assert not _inflight_hazards(b[:2] + [("s_waitcnt", "vmcnt(1)")] + b[2:])
    assert _inflight_hazards(b[:2] + [("s_waitcnt", "vmcnt(2)")] + b[2:])


def test_no_valu_written_sgpr_reaches_vector_memory_within_five_wait_states(release):
    bad = ["%s: %s reads s%s written by a VALU instruction %d wait state(s) earlier" % ((_kernel_name(sym)[:80],) + h)
           for sym, ins in release.items() for h in _sgpr_hazards(ins)]
    assert not bad, "\n".join(bad[:20])


def test_no_register_of_a_load_in_flight_is_touched_before_its_wait(release):
    bad = ["%s: `%s` touches %s while `%s` is in flight" % ((_kernel_name(sym)[:80],) + h)
           for sym, ins in release.items() if any(k in sym for k in HOT) for h in _inflight_hazards(ins)]
    assert not bad, "%d violations, first:\n%s" % (len(bad), "\n".join(bad[:12]))


def _metadata(so, outdir):
    """{kernel symbol: {field: int}} from the code objects' metadata notes"""
    meta = {}
    for co in _code_objects(so, outdir):
        txt = subprocess.run([os.path.join(LLVM, "llvm-readelf"), "--notes", co], capture_output=True, text=True, check=True).stdout
        for blk in txt.split("- .agpr_count:")[1:]:
            name = re.search(r"\.name:\s*(\S+)", blk)
            if not name:
                continue
            get = lambda k: int((re.search(r"\.%s:\s*(\d+)" % k, blk) or [0, "0"])[1])
            meta[name.group(1)] = dict(scratch=get("private_segment_fixed_size"), vgpr_spill=get("vgpr_spill_count"),
                                       sgpr_spill=get("sgpr_spill_count"), vgpr=get("vgpr_count"))
    return meta


def test_hot_kernels_touch_no_scratch_between_their_mfmas(release):
    """Spilled registers inside the MFMA loops were why kernel rows 24 - 27 of the gather-sum table ran up to 13x slower than their
    neighbours (round 3 shipped them as explicit-request variants; they now exist in the tuning build only)."""
    bad, seen = [], 0
    for sym, ins in release.items():
        if not any(h in sym for h in HOT):
            continue
        mf = [i for i, (mn, _) in enumerate(ins) if mn.startswith("v_mfma")]
        if not mf:
            continue
        seen += 1
        inside = [mn for i, (mn, _) in enumerate(ins) if mf[0] < i < mf[-1]]
        scratch = sum(1 for mn in inside if mn.startswith("scratch_"))
        lanes = sum(1 for mn in inside if mn in ("v_readlane_b32", "v_writelane_b32"))
        if scratch or lanes > 8:          # a handful of SGPR reloads (v_readlane) per sweep is tolerated
            bad.append("%s: %d scratch, %d lane moves between the MFMAs" % (_kernel_name(sym)[:100], scratch, lanes))
    assert seen >= 40, seen
    assert not bad, "\n".join(bad)


def test_explicit_request_variants_are_not_in_the_release_library(release):
    names = [_kernel_name(s) for s in release]
    # tile width 32 (rows 24 / 25), three plane buffers (rows 20, 26, 27), three waves per channel (row 7), two stacked images
    # (row 19), the four-wave dense form
    for pat in (r"GatherTraits<1, 1[45], 40, false, 1, 1, 0, 12, 2, 32", r"GatherTraits<[^>]*, 12, 3, 8>", r"GatherTraits<7, 7, 72, true, 3,",
                r"GatherTraits<7, 7, 72, true, 2, 1, 0, 4, 3,", r"GatherTraits<4, 4, 40, false, 2, 2, 13312", r"dense_gather_kernel<\d, 2>"):
        hit = [n for n in names if re.search(pat, n)]
        assert not hit, hit[:3]


def test_release_kernels_are_byte_identical_in_the_tuning_build():
    """The variant tests (tests/util.tuning_capi) run on libdau_conv_hip_tuning.so: what they verify is what ships only if the
    device code is the same.  -DDAU_TUNING changes host code only; the tuning build may hold MORE kernels (explicit-request
    gather variants), never other code for a kernel the release library has."""
    rel, tun = os.path.join(PKG, "libdau_conv_hip.so"), os.path.join(PKG, "libdau_conv_hip_tuning.so")
    if not (os.path.exists(rel) and os.path.exists(tun)):
        pytest.skip("libraries not built")

    def bodies(so, d):
        out = {}
        for co in _code_objects(so, d):
            syms = subprocess.run([os.path.join(LLVM, "llvm-readelf"), "-sW", co], capture_output=True, text=True, check=True).stdout
            sec = subprocess.run([os.path.join(LLVM, "llvm-readelf"), "-SW", co], capture_output=True, text=True, check=True).stdout
            m = re.search(r"\.text\s+PROGBITS\s+([0-9a-f]+)\s+([0-9a-f]+)\s+([0-9a-f]+)", sec)
            addr, off = int(m.group(1), 16), int(m.group(2), 16)
            data = open(co, "rb").read()
            for line in syms.splitlines():
                f = line.split()
                if len(f) >= 8 and f[3] == "FUNC" and f[6].isdigit():
                    a, size = int(f[1], 16), int(f[2])
                    out[f[7]] = data[off + a - addr: off + a - addr + size]
        return out
    with tempfile.TemporaryDirectory() as d:
        r, t = bodies(rel, d), bodies(tun, d)
    assert len(r) > 100
    missing = [_kernel_name(k)[:90] for k in r if k not in t]
    assert not missing, "kernels of the release build that the tuning build lacks: %s" % missing[:5]
    differ = [_kernel_name(k)[:90] for k in r if r[k] != t[k]]
    assert not differ, "%d kernels differ between the release and the tuning build, e.g. %s" % (len(differ), differ[:5])
