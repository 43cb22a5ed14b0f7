cd "$GRAFT_REPO_ROOT" || exit 1
GIT_HEAD=002be0a bash tools/profile_round4.sh r04 "1 5 2 3 4" > gpurun_out/r04_profile.log 2>&1
tail -5 gpurun_out/r04_profile.log
ls gpurun_out/r04/*
