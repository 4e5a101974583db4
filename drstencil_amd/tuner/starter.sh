#!/usr/bin/bash
# starter.sh <stencil.stc> [tuning.py options]  -- counterpart of benchmarks/<stencil>/starter.sh:1-11:
# set up cu/ bin/ prof/, run the search, log the wall time, then scrape metrics of the best configurations.
here=$(cd "$(dirname "$0")" && pwd)
starttime=`date +'%Y-%m-%d %H:%M:%S'`
mkdir -p cu bin prof
cp $here/../csrc/support/common.hpp cu/
python $here/tuning.py "$@" --out tuning_out
endtime=`date +'%Y-%m-%d %H:%M:%S'`
start_seconds=$(date --date="$starttime" +%s)
end_seconds=$(date --date="$endtime" +%s)
echo ${endtime} >> tuning-time.log
echo "running time: "$((end_seconds-start_seconds))"s" >> tuning-time.log
