import re,sys,subprocess
t=open(sys.argv[1]).read()
for e in re.findall(r"error:.*", t): print(e)
blocks=re.split(r"remark: Function Name: ", t)[1:]
pat = sys.argv[2] if len(sys.argv)>2 else 'fused_chain'
for b in blocks:
    name=b.split()[0]
    if pat not in name: continue
    g=lambda k: (re.search(k+r": (\d+)", b) or [None,'?'])[1]
    dn=subprocess.run(['c++filt', name],capture_output=True,text=True).stdout.strip()
    dn=re.sub(r"\(.*","",dn).replace("void wrp::","")
    print("%-52s VGPRs %s spill %s scratch %s SGPRs %s occ %s"%(dn,g(" VGPRs"),g("VGPRs Spill"),g(r"ScratchSize \[bytes/lane\]"),g("TotalSGPRs"),g(r"Occupancy \[waves/SIMD\]")))
