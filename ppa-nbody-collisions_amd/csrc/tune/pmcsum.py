import csv,collections,sys,glob
for d in sys.argv[1:]:
    f=glob.glob("gpurun_out/%s/*counter_collection.csv"%d)+glob.glob("gpurun_out/%s/*/*counter_collection.csv"%d)
    acc=collections.defaultdict(lambda: collections.defaultdict(float)); cnt=collections.Counter()
    for r in csv.DictReader(open(f[0])):
        k=r["Kernel_Name"]
        if "forces" not in k: continue
        acc[k][r["Counter_Name"]]+=float(r["Counter_Value"]); cnt[(k,r["Counter_Name"])]+=1
    for k in acc:
        print(d,k[:60])
        for c,v in sorted(acc[k].items()): print("   %-28s %.4g (n=%d)"%(c,v,cnt[(k,c)]))
