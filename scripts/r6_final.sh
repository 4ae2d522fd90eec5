# GPU box, round 6, last call: the code that ships at the end of the round (code version r6d).  (1) scripts/r6_profile2.sh -- kernel traces and FETCH / WRITE / SQ counter
# passes of the three device workloads; its FETCH / WRITE summaries are copied into the box's profiles/r6/ so that the bench line of step (3) can look its kernels' traffic up
# in passes of THIS code; (2) the whole GPU suite + smoke(); (3) the default bench line.
R=$GRAFT_REPO_ROOT
V=r6d bash $R/scripts/r6_profile2.sh > $R/gpurun_out/r6d_profile.log 2>&1
tail -3 $R/gpurun_out/r6d_profile.log | cut -c1-200
cp $R/gpurun_out/r6d/prof/pmc_fetch_hg38scale_*_r6d.json $R/gpurun_out/r6d/prof/pmc_write_hg38scale_*_r6d.json $R/profiles/r6/
cd $R
bash scripts/r6_call16.sh
