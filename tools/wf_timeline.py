import csv,glob,sys
f=glob.glob(sys.argv[1]+'/*/*kernel_trace.csv')[0]
out=[]
for r in csv.DictReader(open(f)):
    n=r["Kernel_Name"]
    if "k_wf" in n:
        k="T" if "trace" in n else "S" if "shade" in n else "R"
        out.append(f'{k}{(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e6:.1f}')
print(" ".join(out))
