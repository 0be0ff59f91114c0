"""Rollout constants consumed by the loop: same names and values as ``rls/arglist.py:5-24``
(module-level globals that entry scripts mutate, as the reference does at main.py:31-35,64)."""
max_episode_len = 25      # rls/arglist.py:5
num_episodes = 40000      # rls/arglist.py:6
is_training = True        # rls/arglist.py:11
batch_size = 1024         # rls/arglist.py:15
warmup_steps = batch_size  # rls/arglist.py:16
update_rate = 100         # rls/arglist.py:18
display = False           # rls/arglist.py:23
save_rate = 1000          # rls/arglist.py:24
appx = 'scalability/madr/'  # rls/arglist.py:29 (model path prefix used by run_test)
