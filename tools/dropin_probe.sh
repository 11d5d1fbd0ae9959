#!/bin/bash
# where the literal drop-in call's time goes: oracle/_ref/dropin_demo on 1920x1080 camera rays of the bunny scene, per-chunk trace of the pipelined path
python - <<PY
import sys; sys.path.insert(0, ".")
from gravit_amd import scenes
from tests.helpers import oracle_camera_rays
oracle_camera_rays(scenes.bunny_scene(1920, 1080)).tofile("/tmp/rays1080.bin")
PY
for opt in "" "GVT_HIP_ABI_TRACE=1"; do
  env $opt oracle/_ref/dropin_demo tests/golden/bunny.obj /tmp/rays1080.bin 0 /tmp/out.bin 3 2>&1 | tail -40
done
