cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/tl
timeout 900 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $R/gpurun_out/tl/prof -o t -- python3 $R/bench.py --steps 1 --warmup 0 --cpu-sample 0 --throughput-leg 0 --legal-leg 0 --nonideal-leg 0 "$@" > $R/gpurun_out/tl/b.json 2> $R/gpurun_out/tl/b.err
cd $R
python3 - <<'PY'
import csv, glob, re, collections
f=glob.glob('gpurun_out/tl/prof/**/*kernel_trace.csv', recursive=True)[0]
rows=[]
for r in csv.DictReader(open(f)):
    rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'], r.get('Stream_Id', r.get('Queue_Id',''))))
mc=glob.glob('gpurun_out/tl/prof/**/*memory_copy_trace.csv', recursive=True)
for g in mc:
    for r in csv.DictReader(open(g)):
        rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), 'COPY_'+r.get('Direction',r.get('Name','')), 'copy'))
rows.sort()
def short(n):
    m=re.search(r'(\w+)(<[^>(]*>)?\(', n); 
    return (m.group(1)+(m.group(2) or '')) if m else n[:40]
# find DP <1,2> launches as slot markers
marks=[i for i,r in enumerate(rows) if 'ksw_extd2_reg_kernel<1, 2>' in r[2]]
print('slots', len(marks))
# take slots 300..304
out=open('gpurun_out/tl/slots.txt','w')
for k in range(300,304):
    a=rows[marks[k]][0]; b=rows[marks[k+1]][0]
    out.write('--- slot %d: %.3f ms\n'%(k,(b-a)/1e6))
    for r in rows[marks[k]:marks[k+1]]:
        out.write('%9.1f %9.1f %8.1f  %-10s %s\n'%((r[0]-a)/1e3,(r[1]-a)/1e3,(r[1]-r[0])/1e3,r[3][-10:],short(r[2])))
out.close()
# aggregate over slots 200..1000: busy time union and per-kernel totals per slot
agg=collections.defaultdict(float); n=0; busy=0.0; span=0.0
for k in range(200,1000):
    a=rows[marks[k]][0]; b=rows[marks[k+1]][0]; span+=b-a; n+=1
    iv=sorted((r[0],r[1]) for r in rows[marks[k]:marks[k+1]])
    cur_s,cur_e=iv[0]
    for s,e in iv[1:]:
        if s>cur_e: busy+=cur_e-cur_s; cur_s,cur_e=s,e
        else: cur_e=max(cur_e,e)
    busy+=cur_e-cur_s
    for r in rows[marks[k]:marks[k+1]]: agg[short(r[2])]+=r[1]-r[0]
print('mean slot %.3f ms, GPU busy (union) %.3f ms'%(span/n/1e6, busy/n/1e6))
for name,t in sorted(agg.items(), key=lambda x:-x[1])[:25]: print('%-45s %.1f us per slot'%(name[:45], t/n/1e3))
PY
rm -rf gpurun_out/tl/prof
