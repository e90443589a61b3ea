import re,sys,json
for f in sys.argv[1:]:
    txt=open(f).read()
    flips=[int(x) for x in re.findall(r'"flipped_pixels": (\d+)',txt)]
    print(f, "cases",len(flips),"total flips",sum(flips),"max",max(flips), "cases with flips", sum(1 for x in flips if x))
