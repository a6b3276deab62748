import csv,glob,sys,re
f=glob.glob(sys.argv[1]+"/*/*kernel_stats.csv")[0]
for r in csv.DictReader(open(f)):
    n=re.sub(r"\(.*","",r["Name"]).replace("void vxrt::(anonymous namespace)::","").replace("vxrt::(anonymous namespace)::","")
    print(f"{n:40s} calls {r['Calls']:>6s} avg {float(r['AverageNs'])/1e3:9.1f} us  {r['Percentage']}%")
