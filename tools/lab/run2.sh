cd tools/lab
for b in lab2_base lab2_nostore; do for K in 128 256 512 1024; do echo -n "$b K=$K: "; ./$b 262144 512 $K | sed 's/.*: //'; done; done
