import csv, sys, collections
acc=collections.defaultdict(lambda: collections.defaultdict(float)); cnt=collections.Counter()
for d in sys.argv[1:]:
    for r in csv.DictReader(open(d+"/out_counter_collection.csv")):
        k=r["Kernel_Name"].split("(")[0].replace("void ","").replace("motifs::","")[:34]
        acc[k][r["Counter_Name"]]+=float(r["Counter_Value"]); cnt[(k,r["Counter_Name"])]+=1
rows=[]
for k,v in acc.items():
    sal=v.get("SQ_INSTS_SALU",0)/max(cnt[(k,"SQ_INSTS_SALU")],1); val=v.get("SQ_INSTS_VALU",0)/max(cnt[(k,"SQ_INSTS_VALU")],1)
    wc=v.get("SQ_WAVE_CYCLES",0)/max(cnt[(k,"SQ_WAVE_CYCLES")],1); bc=v.get("SQ_BUSY_CYCLES",0)/max(cnt[(k,"SQ_BUSY_CYCLES")],1)
    rows.append((sal*cnt[(k,"SQ_INSTS_SALU")],k,sal,val,bc))
rows.sort(reverse=True)
for t,k,sal,val,bc in rows[:30]:
    # busy cycles: per-SE sum? report SALU per CU-cycle estimate: sal/256 vs kernel cycles (bc/ (num SE=32?))
    print("%-36s SALU/launch %.3g VALU/launch %.3g ratio %.2f  SALU per CU %.3g  BUSY %.3g"%(k,sal,val,sal/max(val,1),sal/256,bc))
