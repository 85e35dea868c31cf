#!/bin/bash
# ONE command that pins this repo's engine against the REAL MuJoCo wherever the wheels install (this container and the GPU box
# have no network: DESIGN.md 6 "parity unpinned"):
#
#     pip install -r tools/pin/requirements.txt && pip install -e /path/to/Metaworld
#     bash tools/pin/run_pin.sh [report.md]
#
# It (1) records the reference's own (seed, actions) -> state / obs / reward / success traces with mujoco + gymnasium + metaworld
# (tools/dump_reference_traces.py -> tests/golden_mujoco/), which also prints the reference's >= 0.8 scripted-policy gate per task
# on the real engine (basketball-v3 FIRST: the one task this repo's engine fails by the reference's own bookkeeping);
# (2) replays them on the host build of the device code (tests/test_mujoco_pin.py; add -m gpu on a GPU box); (3) runs the
# real-gymnasium drop-in test (tests/test_real_gymnasium.py); (4) writes a one-page report.
set -u
cd "$(dirname "$0")/../.."
REPORT=${1:-tools/pin/PIN_REPORT.md}
TMP=$(mktemp -d)
python - <<'PY' > "$TMP/stack.txt" 2>&1
import sys
sys.path.insert(0, ".")
from tools import dump_reference_traces as D
ok, why = D.real_stack_available()
print(("OK: " if ok else "MISSING: ") + why)
sys.exit(0 if ok else 3)
PY
rc=$?
{
  echo "# Engine pin report"
  echo
  echo "- date: $(date -u +%Y-%m-%dT%H:%MZ)"
  echo "- stack: $(cat "$TMP/stack.txt")"
  echo "- device sources: $(python -c 'import sys; sys.path.insert(0, "."); from metaworld_amd import native; print(native.source_hash())')"
} > "$REPORT"
if [ $rc -ne 0 ]; then
  echo "- RESULT: not run -- the real mujoco / gymnasium / metaworld do not import here (pip install -r tools/pin/requirements.txt)" >> "$REPORT"
  cat "$REPORT"; exit 3
fi
python -c 'import __graft_entry__ as g; g.build_host_harness(); from oracle import mjlite; mjlite.build()'
echo; echo "== 1. traces + the reference's scripted-policy gate on the real engine (basketball-v3 first) =="
python tools/dump_reference_traces.py --first basketball-v3 --gate 2>&1 | tee "$TMP/dump.txt"
echo "== 2. does the device code reproduce MuJoCo? (host build, fp64, obs / reward 1e-5 one step from a synchronised state) =="
python -m pytest tests/test_mujoco_pin.py -q -m "not gpu" -p no:cacheprovider 2>&1 | tee "$TMP/pin.txt"
if python -c 'import torch, sys; sys.exit(0 if torch.cuda.is_available() else 1)' 2>/dev/null; then
  python -m pytest tests/test_mujoco_pin.py -q -m gpu -p no:cacheprovider 2>&1 | tee "$TMP/pin_gpu.txt"
fi
echo "== 3. drop-in behind a real gymnasium =="
python -m pytest tests/test_real_gymnasium.py -q -p no:cacheprovider 2>&1 | tee "$TMP/gym.txt"
{
  echo "- traces + gate: $(grep -c '^trace' "$TMP/dump.txt") tasks recorded"
  echo
  echo '```'
  grep -i -E "basketball|gate|success" "$TMP/dump.txt" | head -60
  echo '```'
  echo "- engine vs MuJoCo (host build): $(tail -n 1 "$TMP/pin.txt")"
  [ -f "$TMP/pin_gpu.txt" ] && echo "- engine vs MuJoCo (GPU): $(tail -n 1 "$TMP/pin_gpu.txt")"
  echo "- real gymnasium drop-in: $(tail -n 1 "$TMP/gym.txt")"
  echo
  echo "Failing tasks (if any) with their deviations:"
  echo '```'
  grep -E "^FAILED|AssertionError" "$TMP/pin.txt" | head -60
  echo '```'
} >> "$REPORT"
cat "$REPORT"
