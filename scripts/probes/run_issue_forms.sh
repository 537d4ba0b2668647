set -e
python scripts/probes/issue_forms.py gen > /tmp/issue_forms.hip
hipcc --offload-arch=gfx950 -O3 /tmp/issue_forms.hip -o /tmp/issue_forms 2>/dev/null
timeout -k 10 300 /tmp/issue_forms "$@" > gpurun_out/issue_forms.txt
cat gpurun_out/issue_forms.txt
