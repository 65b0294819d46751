#!/bin/bash
# Round-5 session 25: the forward GEMM with four phases per K-tile against two merged ones -- the kernel's own marks (start, loop start,
# loop end, last store) and the clock it held, cold rows / Infinity-Cache rows / L2-hot rows / no stream.
cd $GRAFT_REPO_ROOT
O=gpurun_out
L=tools/lab/fwd_dr_lab
V="marks_lead_4ph,marks_lead_merged,marks_plain_4ph,marks_plain_merged,marks_lead_4ph_hotA,marks_lead_merged_hotA,marks_lead_4ph_nostream,marks_lead_merged_nostream"
{ echo "== cold rows"; timeout 600 $L 20650 40 2 "$V" 0; echo "== rows in the Infinity Cache"; timeout 600 $L 20650 40 2 "$V" 1; } > $O/r05_s25_marks.txt 2>&1
grep -v "^check" $O/r05_s25_marks.txt | cut -c1-330
