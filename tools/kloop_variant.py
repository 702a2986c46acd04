r"""A/B builds of the Winograd K-loop stream:  python tools/kloop_variant.py NAME KEY=VALUE ... [-- -DMACRO ...]
generates wino_kloop.inc with the generator's environment overrides (KL_*), builds azula_amd/csrc/_ab/libazula_amd_NAME.so
from it (tools/ab_build.py) and restores the default .inc."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
args = sys.argv[1:]
name = args[0]
defs = args[args.index("--") + 1:] if "--" in args else []
kv = [a for a in (args[1:args.index("--")] if "--" in args else args[1:])]
env = dict(os.environ)
env.update(dict(a.split("=", 1) for a in kv))
gen = os.path.join(ROOT, "azula_amd", "csrc", "gen_wino_kloop.py")
try:
    subprocess.run([sys.executable, gen], check=True, env=env, stdout=subprocess.DEVNULL)
    subprocess.run([sys.executable, os.path.join(ROOT, "tools", "ab_build.py"), name, *defs], check=True)
finally:
    clean = {k: v for k, v in os.environ.items() if not k.startswith("KL_")}
    subprocess.run([sys.executable, gen], check=True, env=clean, stdout=subprocess.DEVNULL)
