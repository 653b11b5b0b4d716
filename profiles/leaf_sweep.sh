for L in 1 2 3 4 6 8; do
  echo "LEAF=$L"
  RT_FAST_LEAF=$L timeout -k 10 100 python profiles/sched_stats.py 4 | head -1 | python -c "
import sys,ast
d=ast.literal_eval(sys.stdin.read())
print('  node/ray %.2f sphere/ray %.2f fallbacks %d' % (d['node_tests']/d['rays'], d['sphere_tests']/d['rays'], d['fallbacks']))"
  RT_FAST_LEAF=$L bash profiles/sweep.sh "RT_BOX_KEEP=8"
done
