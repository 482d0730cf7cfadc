import sys, numpy as np
a=np.loadtxt(sys.argv[1],dtype=np.int64); a=a[(a[:,2]!=0)|(a[:,0]!=0)]
u32=1<<32
tail=(a[:,2]-a[:,1])%u32; life=(a[:,2]-a[:,0])%u32
trips=a[:,3]&4095; ev=(a[:,3]>>12)&1023; ml=((a[:,3]>>22)&1023)/16.0
q=lambda v:" ".join(f"{np.percentile(v,p):.0f}" for p in (10,50,90,100))
print(sys.argv[2],"tail clocks p10/50/90/max",q(tail),"| trips after dry",q(trips),"| event phases after dry",q(ev),"| lanes per trip after dry",q(ml))
p=trips+ev; ok=p>0
print(sys.argv[2],"clocks per pass after dry (median over waves)", np.median(tail[ok]/p[ok]), " corr(tail, passes)", np.corrcoef(tail[ok],p[ok])[0,1])
