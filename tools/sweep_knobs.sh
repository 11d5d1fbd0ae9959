# usage (GPU box): bash tools/sweep_knobs.sh "opt=value ..." "opt=value ..."   -- tools/tail_probe.py per option set, 1 M-ray rows + fits
for o in "$@"; do echo "== $o"; timeout -k 10 200 python tools/tail_probe.py any $o 2>&1 | grep -E "1032256|1003466|fit"; done
