"""CPU-side audit of the built gfx950 code objects inside libwarprnnt.so (no GPU needed): the hot kernels keep their
state in registers (no scratch), use the instructions DESIGN.md says they use, and stay within the register budgets
their occupancy assumptions rest on.  Guards against silent regressions (a spill in a sweep or MFMA loop costs far more
than it looks) that parity tests cannot see."""
import os
import re
import shutil
import struct
import subprocess
import tempfile

import pytest

import rnnt_speech_recognition_amd as pkg

LLVM = "/opt/rocm/lib/llvm/bin"
READELF, OBJDUMP = os.path.join(LLVM, "llvm-readelf"), os.path.join(LLVM, "llvm-objdump")
pytestmark = pytest.mark.skipif(not (os.path.exists(READELF) and os.path.exists(OBJDUMP)), reason="ROCm LLVM tools absent")


def _code_objects(so_path, out_dir):
    """Pull the gfx950 ELF images out of the clang offload bundles embedded in .hip_fatbin."""
    data = open(so_path, "rb").read()
    magic = b"__CLANG_OFFLOAD_BUNDLE__"
    paths = []
    for m in re.finditer(magic, data):
        base = m.start()
        (n,) = struct.unpack_from("<Q", data, base + len(magic))
        off = base + len(magic) + 8
        for _ in range(n):
            eoff, esize, tlen = struct.unpack_from("<QQQ", data, off)
            triple = data[off + 24 : off + 24 + tlen].decode()
            off += 24 + tlen
            if "amdgcn" in triple and "gfx950" in triple and esize:
                p = os.path.join(out_dir, f"co{len(paths)}.elf")
                open(p, "wb").write(data[base + eoff : base + eoff + esize])
                paths.append(p)
    return paths


@pytest.fixture(scope="module")
def kernels():
    so = pkg.build()
    tmp = tempfile.mkdtemp(prefix="isa_audit_")
    try:
        meta, asm = {}, {}
        for co in _code_objects(so, tmp):
            notes = subprocess.run([READELF, "--notes", co], capture_output=True, text=True).stdout
            cur = {}
            for line in notes.splitlines():
                mm = re.match(r"\s*-?\s*\.(\w+):\s*(.*)$", line)
                if not mm:
                    continue
                k, v = mm.group(1), mm.group(2).strip().strip("'")
                cur[k] = v  # (argument records repeat some keys; the kernel-level ones come last and win)
                if k == "wavefront_size":  # last key of a kernel record in the msgpack dump
                    if "symbol" in cur:
                        meta[cur["symbol"]] = dict(cur)
                    cur = {}
            dis = subprocess.run([OBJDUMP, "-d", "--no-show-raw-insn", co], capture_output=True, text=True).stdout
            name = None
            for line in dis.splitlines():
                mm = re.match(r"^[0-9a-f]+ <(.+)>:$", line)
                if mm:
                    name = mm.group(1)
                    asm[name] = []
                elif name is not None:
                    asm[name].append(line)
        yield meta, {k: "\n".join(v) for k, v in asm.items()}
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def _find(d, *needles):
    hits = [k for k in d if all(n in k for n in needles)]
    assert hits, needles
    return hits


def test_code_objects_are_gfx950_and_have_the_kernels(kernels):
    meta, asm = kernels
    assert len(meta) > 20
    for needles in (("sweep_ld_kernel", "Li3ELi16E"), ("cell_tile_kernel", "Li32ELb1"), ("cell_tile_kernel", "Li32ELb0"),
                    ("jh_logits_kernel", "Li40ELi0"), ("jh_logits_kernel", "Li40ELi1"), ("jh_logits_kernel", "Li40ELi2"),
                    ("jh_dhx_kernel", "Li5E"), ("jh_dhx_kernel", "Li1E"), ("jh_dw_kernel",),
                    ("lin_sweep_kernel", "Li3ELi16E"), ("lin_redo_kernel", "Li3ELi16E"), ("joint_redo_kernel", "Li3ELi16E"),
                    ("joint_fwd_kernel",), ("joint_bwd_kernel",), ("joint_cellrec_kernel",), ("joint_rowplan_kernel",), ("joint_reduce_kernel",),
                    ("dense_gemm_nt_kernel",), ("dense_gemm_tn_kernel",)):
        _find(meta, *needles)


def test_hot_kernels_do_not_spill(kernels):
    meta, _ = kernels
    hot = [("sweep_ld_kernel",), ("cell_tile_kernel", "Li32ELb1ELb1"),
           ("cell_tile_kernel", "Li32ELb0ELb1"), ("jh_logits_kernel", "Li40ELi0"), ("jh_dw_kernel", "Li512E"),
           ("joint_dl_kernel",), ("joint_phase1s_kernel",),
           # round 4 / 5: the linear-domain sweeps, both hand-back kernels, the fused f32-grade joint and its first Dense layer
           ("lin_sweep_kernel",), ("joint_redo_kernel",), ("joint_fwd_kernel",), ("joint_bwd_kernel",),
           ("joint_cellrec_kernel",), ("joint_rowplan_kernel",), ("joint_reduce_kernel",), ("dense_gemm_nt_kernel",), ("dense_gemm_tn_kernel",)]
    for needles in hot:
        for k in _find(meta, *needles):
            m = meta[k]
            assert int(m["private_segment_fixed_size"]) == 0, (k, m["private_segment_fixed_size"])
            assert int(m.get("vgpr_spill_count", "0")) == 0, k
    # the wide joint's phase 2 (640 < J <= 704, two 4-wave workgroups per CU = 256 registers; this file is compiled without the SLP
    # vectoriser since round 5 -- packed f32 does not overlap with the matrix pipe -- which costs this cold kernel two spilled registers)
    for k in _find(meta, "joint_phase2s_kernel"):
        assert int(meta[k]["private_segment_fixed_size"]) <= 16, (k, meta[k]["private_segment_fixed_size"])
    # the loss op's hand-back kernel (cold path; 1024 threads = 128 registers; a gradient row of up to 60 symbols per lane next to
    # six 16-byte pieces in flight): a handful of spilled registers are tolerated
    for k in _find(meta, "lin_redo_kernel"):
        assert int(meta[k]["private_segment_fixed_size"]) <= 64, (k, meta[k]["private_segment_fixed_size"])
    # the dW2 kernel's 256-column instantiations (vocabularies of 128 / 256 symbols) are compiled for FOUR waves per SIMD -- two
    # workgroups per CU, measured 10 % faster than one -- and pay for the 128-register budget with a few spilled registers
    for k in _find(meta, "jh_dw_kernel", "Li256E"):  # (round 6: + the unit's row list and the scalar row look-ups: 88 bytes)
        assert int(meta[k]["vgpr_count"]) <= 128 and int(meta[k]["private_segment_fixed_size"]) <= 96, (k, meta[k])
    # round 5: V = 128 has its own 128-column instantiation (before: the half-empty 256-column one with 71 spilled registers)
    for k in _find(meta, "jh_dw_kernel", "Li128E"):
        assert int(meta[k]["vgpr_count"]) <= 128 and int(meta[k]["private_segment_fixed_size"]) == 0, (k, meta[k])
    assert not [k for k in meta if "jh_dw_kernel" in k and "Lb1ELi256E" in k], "the partial 256-column dW2 kernel is gone"
    # round 6: the fused dlogits + dh kernel (32 NT accumulators + the fragments fill the 256 registers at J = 640): a few spilled
    # registers in its per-iteration EPILOGUE are tolerated, none in the main loop -- a scratch reload there waits for every LDS-DMA
    # piece in flight (test_dhx_main_loop_has_no_scratch_and_counted_waits)
    for k in _find(meta, "jh_dhx_kernel"):
        assert int(meta[k]["private_segment_fixed_size"]) <= 64, (k, meta[k]["private_segment_fixed_size"])  # (0 in the tree as committed)
    # the logits kernels with a [cells][V] epilogue (park / recompute) at J = 640 are allowed a handful of spilled registers
    for k in _find(meta, "jh_logits_kernel", "Li40ELi1") + _find(meta, "jh_logits_kernel", "Li40ELi2"):
        assert int(meta[k]["private_segment_fixed_size"]) <= 64, meta[k]["private_segment_fixed_size"]


def test_register_budgets_match_the_occupancy_assumptions(kernels):
    meta, _ = kernels
    # 8-wave workgroups of the f16 joint: two waves per SIMD -> at most 256 unified registers per wave
    for needles in (("jh_logits_kernel", "Li40"), ("jh_dhx_kernel",), ("jh_dw_kernel",)):
        for k in _find(meta, *needles):
            assert int(meta[k]["vgpr_count"]) + int(meta[k].get("agpr_count", "0")) <= 256, (k, meta[k]["vgpr_count"])
    # split-precision phase 2: two 4-wave workgroups per CU -> at most 256 registers
    for k in _find(meta, "joint_phase2s_kernel"):
        assert int(meta[k]["vgpr_count"]) + int(meta[k].get("agpr_count", "0")) <= 256, (k, meta[k]["vgpr_count"])
    # lane-per-cell patch kernels: six 26.9 KB workgroups per CU at V = 28 = 24 waves -> at most 80 registers for full residency
    for k in _find(meta, "cell_tile_kernel", "Li32"):
        assert int(meta[k]["vgpr_count"]) <= 80, (k, meta[k]["vgpr_count"])


def test_instruction_selection(kernels):
    _, asm = kernels
    for mode in ("Li40ELi0", "Li40ELi1"):  # forward, forward + park
        k1 = asm[_find(asm, "jh_logits_kernel", mode)[0]]
        # (binary16 operands by fused multiply-add + convert -- v_fma_mixlo/hi_f16 -- since the file is compiled without the SLP
        # vectoriser; before: v_pk_fma_f32 + v_cvt_pk_f16_f32)
        assert k1.count("v_mfma_f32_32x32x16_f16") >= 40 and "global_load_lds_dwordx4" in k1 and ("v_fma_mixlo_f16" in k1 or "v_cvt_pk_f16_f32" in k1)
    dw = asm[_find(asm, "jh_dw_kernel")[0]]
    assert "ds_read_b64_tr_b16" in dw and "v_mfma_f32_32x32x16_f16" in dw and "v_dot2" in dw
    dh = asm[_find(asm, "jh_dhx_kernel", "Li5E")[0]]  # round 6: dlogits from the parked values + dh in one kernel, both operands by LDS-DMA
    assert "global_load_lds_dwordx4" in dh and "global_load_lds_dword " in dh and dh.count("v_mfma_f32_32x32x16_f16") >= 80
    assert "global_store_dwordx4" in dh and " nt" in dh  # the converted rows go back to dl for the dW2 kernel, once
    ld = asm[_find(asm, "sweep_ld_kernel", "Li3ELi16E")[0]]  # the default sweep: sweeping wave + loader wave
    assert "global_load_lds_dwordx4" in ld and "global_store_dwordx3" in ld and "v_pk_add_f32" in ld
    assert ld.count("s_barrier") == 1                 # only the counter-initialisation barrier; the hand-off is two LDS counters
    assert "wave_shr:1" in ld and "wave_shl:1" in ld  # ONE whole-wave DPP shift per diagonal carries the neighbour column
    assert ld.count("v_exp_f32") >= 96 and ld.count("v_log_f32") >= 96  # 2 directions x 16 unrolled diagonals x 3 columns
    # the linear-domain sweeps (the loss op's default and, since round 5, the fused f32-grade joint's): multiply / add only on the
    # serial chain -- no transcendental anywhere in the kernel --, one whole-wave DPP shift per unrolled diagonal and direction
    for kk, g in ((1, 16), (2, 16), (3, 16), (4, 16), (6, 8), (8, 8), (12, 4), (16, 4)):
        ln = asm[_find(asm, "lin_sweep_kernel", f"Li{kk}ELi{g}E")[0]]
        assert "v_exp_f32" not in ln and "v_log_f32" not in ln and "v_rcp_f32" not in ln, kk
        assert "global_load_lds_dwordx4" in ln and "v_ldexp_f32" in ln and "v_frexp_exp_i32_f32" in ln
        assert ln.count("wave_shr:1") >= g and ln.count("wave_shl:1") >= g, kk  # (the renormalisation's look-back shifts come on top)
    l3 = asm[_find(asm, "lin_sweep_kernel", "Li3ELi16E")[0]]
    assert "global_store_dwordx3" in l3 and l3.count("s_barrier") == 1
    for name in ("lin_redo_kernel", "joint_redo_kernel"):  # the log-domain redo: float64 recurrence, LDS-DMA loader, agent-scope phase counter
        rd = asm[_find(asm, name, "Li3ELi16E")[0]]
        assert "v_add_f64" in rd and "global_load_lds_dwordx4" in rd and "global_atomic_add" in rd
    # f32-parity joint: split-precision products (hi + lo binary16 operands) on the f16 MFMA units, in every kernel of it
    for name, n_mfma in (("joint_fwd_kernel", 6), ("joint_bwd_kernel", 12), ("joint_phase1s_kernel", 12), ("joint_phase2s_kernel", 12),
                         ("dense_gemm_nt_kernel", 12), ("dense_gemm_tn_kernel", 12)):
        ks = asm[_find(asm, name)[0]]
        assert ks.count("v_mfma_f32_32x32x16_f16") >= n_mfma and "v_mfma_f32_32x32x2_f32" not in ks, name
    for name in ("joint_fwd_kernel", "joint_bwd_kernel", "joint_phase1s_kernel", "joint_phase2s_kernel"):
        assert "v_cvt_pk_f16_f32" in asm[_find(asm, name)[0]] and "v_fma_mix_f32" in asm[_find(asm, name)[0]], name
    fw = asm[_find(asm, "joint_fwd_kernel")[0]]
    assert "v_permlane32_swap" in fw and "row_newbcast" in fw and "global_load_lds_dwordx4" in fw and "s_barrier" in fw
    assert "ds_read_b64_tr_b16" in asm[_find(asm, "dense_gemm_tn_kernel")[0]]
    for name, text in asm.items():
        assert "v_mfma_f32_32x32x2_f32" not in text, name  # round 5: no plain-f32 MFMA fallback left (W2 is scaled into binary16's range)
    # round 5: no packed-f32 arithmetic in the main loops of the MFMA kernels (a v_pk_fma_f32 does not overlap with the matrix pipe:
    # scripts/probes/probe_pk.hip); the 64-bit DPP broadcast of the forward kernel does
    for name in ("joint_bwd_kernel", "jh_dhx_kernel", "jh_dw_kernel", "dense_gemm_nt_kernel", "dense_gemm_tn_kernel"):
        for k in _find(asm, name):
            assert not re.search(r"v_pk_(fma|mul|add)_f32", asm[k]), k
    assert "v_mov_b64_dpp" in fw
    # round 5 (late): a producer of joint_bwd_kernel evaluates a row's dlogits ONCE (16 exponentials per lane; the slow-tanh consumer
    # instance has 16 more) and gathers the dW2 operand from the LDS image it has just written: 32 zero-extending 16-bit reads per
    # lane -- not d16 loads into register halves, which clear the other half with SRAM ECC on (every MI300 / MI355)
    bw = asm[_find(asm, "joint_bwd_kernel")[0]]
    assert len(re.findall(r"\bv_exp_f32", bw)) == 32, len(re.findall(r"\bv_exp_f32", bw))
    assert len(re.findall(r"\bds_read_u16 ", bw)) == 32 and "ds_read_u16_d16" not in bw
    for name, text in asm.items():
        assert "v_mfma_f32_32x32x8" not in text  # no CDNA3-shaped f16 MFMAs: gfx950 forms only


def test_no_transcendental_result_is_read_by_the_next_instruction(kernels):
    """gfx950 needs one wait state between a transcendental (v_rcp / v_exp / v_log / v_rsq / v_sqrt / v_sin / v_cos) and a
    VALU instruction that reads its result.  The compiler inserts it for the instructions it can see -- not for inline asm
    (a v_cvt_pk_f16_f32 written in asm once read the stale 1 + e^x instead of its reciprocal: inf -> NaN costs).  Every
    kernel of the library is scanned: the destination of a transcendental must not be a source of the next instruction."""
    _, asm = kernels
    trans = re.compile(r"^\s*(v_(?:rcp|exp|log|rsq|sqrt|sin|cos)_(?:f32|f16|legacy_f32|iflag_f32))\S*\s+(v\d+)\b")
    bad = []
    for name, text in asm.items():
        ins = [l.split("//")[0].rstrip() for l in text.splitlines() if re.match(r"^\s+[a-z]", l)]
        for cur, nxt in zip(ins, ins[1:]):
            m = trans.match(cur)
            if not m or not re.match(r"^\s*v_", nxt):
                continue
            dst = int(m.group(2)[1:])
            ops = nxt.split(None, 1)[1] if len(nxt.split(None, 1)) > 1 else ""
            srcs = ops.split(",", 1)[1] if "," in ops else ""
            if nxt.lstrip().startswith(("v_fma_mixhi", "v_fma_mixlo", "v_fmac", "v_mac", "v_dot2c", "v_mfma")):
                srcs = ops  # these read their destination too
            regs = set()
            for lo, hi in re.findall(r"v\[(\d+):(\d+)\]", srcs):
                regs.update(range(int(lo), int(hi) + 1))
            regs.update(int(r) for r in re.findall(r"\bv(\d+)\b", re.sub(r"v\[\d+:\d+\]", "", srcs)))
            if dst in regs:
                bad.append((name[:60], cur.strip(), nxt.strip()))
    assert not bad, bad[:5]


def _vregs(tok):
    tok = tok.strip().split(" ")[0]
    m = re.match(r"^v(\d+)$", tok)
    if m:
        return {int(m.group(1))}
    m = re.match(r"^v\[(\d+):(\d+)\]$", tok)
    return set(range(int(m.group(1)), int(m.group(2)) + 1)) if m else set()


def _lds_results_read_before_their_wait(text):
    """Straight-line scan of one kernel's disassembly: every ds_read / ds_bpermute ... destination register stays "pending" until an
    s_waitcnt lgkmcnt(N) retires it (LDS returns in order: all but the newest N operations; with a scalar load in flight only
    lgkmcnt(0) counts), a later instruction overwrites it, or a branch ends the block.  Returns the instructions that read one."""
    bad, pending, issued, smem = [], [], 0, False
    for line in text.splitlines():
        ins = line.split("//")[0].strip()
        if not ins or not re.match(r"^[a-z]", ins):
            continue
        parts = ins.split(None, 1)
        op, ops = parts[0], [o.strip() for o in (parts[1].split(",") if len(parts) > 1 else [])]
        if op.startswith(("s_cbranch", "s_branch", "s_setpc", "s_endpgm")):
            pending = []
            continue
        if op == "s_waitcnt":
            m = re.search(r"lgkmcnt\((\d+)\)", ins)
            if m:
                n = int(m.group(1))
                if n == 0:
                    pending, smem = [], False
                elif not smem:
                    pending = [p for p in pending if p[0] > issued - n]
            continue
        lds_read = op.startswith(("ds_read", "ds_bpermute", "ds_permute", "ds_swizzle")) or (op.startswith("ds_") and "rtn" in op)
        has_dst = lds_read or op.startswith(("v_", "global_load", "buffer_load"))
        used = set()
        for o in (ops[1:] if has_dst else ops):
            used |= _vregs(o)
        if any(used & rs for _, rs in pending):
            bad.append(ins)
        if has_dst and ops:
            w = _vregs(ops[1]) if (op.startswith("v_") and ops[0].startswith(("s", "vcc")) and len(ops) > 1) else _vregs(ops[0])
            pending = [(q, rs - w) for q, rs in pending if rs - w]
        if op.startswith("ds_"):
            issued += 1
            if lds_read:
                pending.append((issued, _vregs(ops[0])))
        elif op.startswith(("s_load", "s_buffer_load", "s_memtime", "s_memrealtime")):
            issued, smem = issued + 1, True
    return bad


def test_no_lds_read_result_is_used_before_its_wait(kernels):
    """The compiler counts waits for the loads IT emits; an LDS read written in inline asm is invisible to it, and a consumer of
    the asm's output may be scheduled in front of a separate `s_waitcnt` asm (round 5: four v_lshl_or_b32 of joint_bwd_kernel's
    producer sat in front of theirs -- right by luck, the reads were ~130 cycles old).  Every kernel of the library is scanned."""
    _, asm = kernels
    probe = "ds_read_u16 v5, v4\nv_lshl_or_b32 v6, v5, 16, v7\ns_waitcnt lgkmcnt(0)\nv_mov_b32_e32 v8, v5\n"
    assert _lds_results_read_before_their_wait(probe) == ["v_lshl_or_b32 v6, v5, 16, v7"]  # (the checker sees what it is for)
    bad = {name: found[:3] for name, text in asm.items() if (found := _lds_results_read_before_their_wait(text))}
    assert not bad, bad


def test_dhx_main_loop_has_no_scratch_and_counted_waits(kernels):
    """jh_dhx_kernel keeps three W2 stages and four A stages in flight by LDS-DMA; anything the compiler tracks on the vector-memory
    counter inside the main loop -- a scratch reload, an ordinary load -- is waited for with vmcnt(0), i.e. for every piece in
    flight (measured: the kernel's step time doubled).  The four unrolled steps of the loop are the regions between the first five
    barriers that are followed by MFMAs."""
    _, asm = kernels
    for k in _find(asm, "jh_dhx_kernel"):
        regions = re.split(r"\bs_barrier\b", asm[k])
        steps = [r for r in regions if r.count("v_mfma_f32_32x32x16_f16") >= 4]
        assert len(steps) >= 4, (k, len(steps))
        for r in steps[:3]:  # (the fourth region runs on into the epilogue)
            assert "scratch_" not in r, k
            assert not re.search(r"\bglobal_load_dword", r.replace("global_load_lds_dword", "")), k
