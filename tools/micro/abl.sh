cd tools/micro
run() { for ab in 15 31; do echo -n "$1 cfg $2 ablate $ab: "; PW_CFG=$2 PW_ABLATE=$ab timeout -k 10 100 ./pw_bench 256 "$1" | grep "pw cfg" | awk '{print $4, $5}'; done; }
run s3.conv3 7
run s3.conv3 9
run "s2.conv3" 7
run "s1.conv3" 7
run "s1.conv3" 6
