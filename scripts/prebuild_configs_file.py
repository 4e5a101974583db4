import os,sys
sys.path.insert(0,'/root/repo')
from concurrent.futures import ProcessPoolExecutor
from drstencil_amd.tuner import tuning as t
stc=os.path.abspath(sys.argv[2] if len(sys.argv)>2 else '/root/repo/benchmarks/configs/c4_3d7pt_star_1024.stc')
jobs=[]
for l in open(sys.argv[1]):
    l=l.strip()
    if not l or l.startswith('#'): continue
    jobs.append((l.replace(" ","").replace("--","_"), ["--3d","--dtype","fp32","--cc-opt","-fno-slp-vectorize"]+l.split()+[stc]))
with ProcessPoolExecutor(max_workers=8) as ex:
    res=list(ex.map(t._build, jobs, chunksize=2))
print(len(jobs), sum(1 for r in res if r[1]), "built")
