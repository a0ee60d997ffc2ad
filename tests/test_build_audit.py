"""The build's per-kernel resource audit (`quantool_amd/csrc/build.py:_resources`): hipcc's kernel-resource-usage remarks
are parsed into `csrc/_obj/<file>.resources.txt`, and a kernel with a private segment (scratch) fails the build.  Round 4
shipped -- for a few commits -- every `sgemm_tn_kernel` variant with its argument struct in scratch; nothing but this
audit can see that without a GPU."""
import re

import pytest

from quantool_amd.csrc import build as qbuild

REMARKS = """\
/x/sgemm_tn.hip:20:1: remark: Function Name: _ZN3foo6kernelE [-Rpass-analysis=kernel-resource-usage]
   20 | __global__ void kernel()
      | ^
/x/sgemm_tn.hip:20:1: remark:     TotalSGPRs: 43 [-Rpass-analysis=kernel-resource-usage]
/x/sgemm_tn.hip:20:1: remark:     VGPRs: 96 [-Rpass-analysis=kernel-resource-usage]
/x/sgemm_tn.hip:20:1: remark:     AGPRs: 0 [-Rpass-analysis=kernel-resource-usage]
/x/sgemm_tn.hip:20:1: remark:     ScratchSize [bytes/lane]: SCRATCH [-Rpass-analysis=kernel-resource-usage]
/x/sgemm_tn.hip:20:1: remark:     Dynamic Stack: False [-Rpass-analysis=kernel-resource-usage]
/x/sgemm_tn.hip:20:1: remark:     Occupancy [waves/SIMD]: 5 [-Rpass-analysis=kernel-resource-usage]
/x/sgemm_tn.hip:20:1: remark:     SGPRs Spill: 0 [-Rpass-analysis=kernel-resource-usage]
/x/sgemm_tn.hip:20:1: remark:     VGPRs Spill: 0 [-Rpass-analysis=kernel-resource-usage]
/x/sgemm_tn.hip:20:1: remark:     LDS Size [bytes/block]: 16384 [-Rpass-analysis=kernel-resource-usage]
/x/sgemm_tn.hip:99:7: warning: unused variable 'q' [-Wunused-variable]
"""


def test_remarks_become_a_table_and_real_warnings_pass_through(tmp_path, monkeypatch):
    monkeypatch.setattr(qbuild, "OBJ_DIR", tmp_path)
    rest = qbuild._resources(tmp_path / "sgemm_tn.hip", REMARKS.replace("SCRATCH", "0"))
    assert "unused variable" in rest and "kernel-resource-usage" not in rest
    row = (tmp_path / "sgemm_tn.resources.txt").read_text().strip().split("\t")
    assert row[0] == "_ZN3foo6kernelE" and "vgpr 96" in row and "scratch 0" in row and "lds 16384" in row


def test_a_kernel_with_scratch_fails_the_build(tmp_path, monkeypatch):
    monkeypatch.setattr(qbuild, "OBJ_DIR", tmp_path)
    with pytest.raises(RuntimeError, match="private segment"):
        qbuild._resources(tmp_path / "sgemm_tn.hip", REMARKS.replace("SCRATCH", "264"))


def test_the_built_library_has_no_kernel_with_scratch():
    """What the last build wrote (build() ran before the tests): every kernel of every translation unit at 0 bytes."""
    tables = sorted(qbuild.OBJ_DIR.glob("*.resources.txt"))
    if not tables:
        pytest.skip("no build in this tree yet")
    assert {t.name.split(".")[0] for t in tables} >= {p.stem for p in qbuild.sources() if p.stem != "capi"}
    n = 0
    for t in tables:
        for line in t.read_text().splitlines():
            n += 1
            m = re.search(r"\tscratch (\d+)\t", line)
            assert m and int(m.group(1)) == 0, line
    assert n >= 60
