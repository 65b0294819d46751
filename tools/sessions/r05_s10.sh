#!/bin/bash
# Round-5 session 10: the phase's LDS-DMA instructions in front of its fragment reads (both GEMMs, lab instantiations): bit-identity and time.
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out
timeout 600 tools/lab/fwd_dr_lab 20650 40 3 "ph_plain,ph_sf_plain,ph_sf_lead,ph_lead2" 0 > $O/r05_s10_fwd_sf_cold.txt 2>&1
timeout 600 tools/lab/fwd_dr_lab 20650 40 2 "ph_plain,ph_sf_plain,ph_sf_lead,ph_lead2" 1 > $O/r05_s10_fwd_sf_ic.txt 2>&1
timeout 600 tools/lab/wgrad_ts_lab 20650 40 3 0 > $O/r05_s10_wgrad_sf_cold.txt 2>&1
timeout 600 tools/lab/wgrad_ts_lab 20650 40 2 1 > $O/r05_s10_wgrad_sf_ic.txt 2>&1
grep -E "^check|^round" $O/r05_s10_fwd_sf_cold.txt; grep -E "^round" $O/r05_s10_fwd_sf_ic.txt
grep -E "^check|^round|wgrad_ts_sf" -A2 $O/r05_s10_wgrad_sf_cold.txt | grep -E "^check|^round  *[0-9]  *wgrad  |wgrad_sf|wgrad_ts_sf|stamps" | cut -c1-330 | head -60
grep -E "^round" $O/r05_s10_wgrad_sf_ic.txt | grep -E "wgrad  |wgrad_sf"
