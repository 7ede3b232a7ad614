import csv,glob,sys,collections
f=glob.glob(sys.argv[1]+"/**/*_counter_collection.csv",recursive=True)[0]
acc=collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f)):
    k=r["Kernel_Name"]
    if "wp3" in k or "conv_wgrad_kernel" in k or "conv_p3" in k or "conv_p8" in k or "conv_wp8" in k or "conv_igemm" in k:
        acc[k[:72]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k,v in acc.items():
    print(k)
    for c,x in v.items(): print("   %-28s %14.0f  (n=%d)"%(c,sum(x)/len(x),len(x)))
