cd $GRAFT_REPO_ROOT
tag=r04
bash scripts/collect_profiles.sh ${tag} && \
bash scripts/collect_profiles.sh ${tag}_k1024 --topics 1024 --steps 10 --warmup 2 --no-cpu-baseline && \
bash scripts/collect_profiles.sh ${tag}_c4 --docs 18846 --types 60000 --mean-len 150 --topics 200 --no-cpu-baseline
