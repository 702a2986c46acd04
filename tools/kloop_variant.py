r"""A/B builds of the hand-scheduled K-loop streams:  python tools/kloop_variant.py NAME KEY=VALUE ...
regenerates wino_kloop.inc / igemm_kloop.inc with the generators' overrides (KL_* / KG_*, honoured because AZ_KLOOP_AB=1 is set
here) inside a COPY of azula_amd/csrc -- the committed .inc files are never touched -- and builds
azula_amd/csrc/_ab/libazula_amd_NAME.so from it (tools/ablate.py; select with AZULA_AMD_LIB=<path>)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import ablate  # noqa: E402

name, kv = sys.argv[1], sys.argv[2:]
ablate.B.build()
print(ablate.build(name, patches=[], regen_env=dict(a.split("=", 1) for a in kv)))
