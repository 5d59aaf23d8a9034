import sys,re
for l in sys.stdin:
    if 'IdctTile' in l or 'FusedGab' in l or 'HfDecodeSimt' in l or 'LfDecodeKernel' in l or 'timed region' in l:
        print(l.rstrip())
