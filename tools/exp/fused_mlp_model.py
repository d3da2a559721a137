"""Work-list model of a fused FFN1 + FFN2 persistent launch: 32 CUs of one XCD draw items in list order; an FFN1 item =
19.4 us of K loop + a 256-unit epilogue burst that shares the HBM (5600 units/us chip-wide, 46 per CU) with every CU bursting at the time (x 8 XCDs in step);
an FFN2 item = 58 us of K loop + a small burst and waits for the 12 FFN1 items of its row panel.  Prints the makespan of several list orders."""
def sim(order, ncu=32, nx=8, BW=5.6e3, rcu=46.0, dt=0.05, G1=(19.4,256.0), G2=(58.0,48.0), pre=False):
    done1={p:12 for p in range(8)} if pre else {}
    q=list(order); qi=0
    cus=[None]*ncu
    T=0.0; finished=0; total=len(q)
    while finished<total and T<2000:
        for c in range(ncu):
            if cus[c] is None and qi<len(q):
                kind,p=q[qi]; qi+=1
                k,e=(G1 if kind=='1' else G2)
                cus[c]=[kind,p,'w' if kind=='2' else 'k',k,e]
        nb=sum(1 for x in cus if x and x[2]=='e')*nx
        rate=min(rcu,BW/nb) if nb else 0
        for c in range(ncu):
            x=cus[c]
            if not x: continue
            if x[2]=='w':
                if done1.get(x[1],0)>=12: x[2]='k'
                else: continue
            if x[2]=='k':
                x[3]-=dt
                if x[3]<=0: x[2]='e'
            elif x[2]=='e':
                x[4]-=rate*dt
                if x[4]<=0:
                    if x[0]=='1': done1[x[1]]=done1.get(x[1],0)+1
                    cus[c]=None; finished+=1
        T+=dt
    return round(T,1)
P=range(8)
allg1=[('1',p) for p in P for _ in range(12)]
allg2=[('2',p) for p in P for _ in range(4)]
print("separate launches (+4 launch/prologue):", sim(allg1)+sim(allg2,pre=True)+4)
print("all G1 then all G2 in one list:", sim(allg1+allg2))
def inter(lag):
    o=[]
    for p in P:
        o+= [('1',p)]*12
        if p-lag>=0: o+=[('2',p-lag)]*4
    for p in range(8-lag,8): o+=[('2',p)]*4
    return o
for lag in (1,2,3,4): print("interleaved lag",lag, sim(inter(lag)))
o=[('1',0)]*12+[('1',1)]*12+[('1',2)]*12+[('2',0)]*4+[('1',3)]*12+[('1',4)]*12+[('2',1)]*4+[('1',5)]*12+[('1',6)]*12+[('2',2)]*4+[('1',7)]*12
for p in range(3,8): o+=[('2',p)]*4
print("front-loaded G1:", sim(o))
o=[('1',0)]*12+[('1',1)]*12+[('2',0)]*4+[('1',2)]*12+[('2',1)]*4+[('1',3)]*12+[('1',4)]*12+[('2',2)]*4+[('1',5)]*12+[('1',6)]*12+[('1',7)]*12
for p in range(3,8): o+=[('2',p)]*4
print("variant b:", sim(o))
# G2 as 8 items per panel of half the rows?? (128-row tiles at same efficiency, hypothetical): 64 items x 29
o2=inter(2); print("lag 2 with G2 in halves (hypothetical 29us items):", sim([(k,p) for (k,p) in o2 for _ in range(2 if k=='2' else 1)], G2=(29.0,24.0)))
