import glob, os
for c in sorted(glob.glob('/sys/class/drm/card*/device')):
    print(c, os.path.exists(c + '/pp_dpm_sclk'))
    for h in glob.glob(c + '/hwmon/hwmon*'):
        for f in ('freq1_input', 'freq1_label', 'power1_average', 'power1_input', 'power1_cap'):
            p = os.path.join(h, f)
            if os.path.exists(p):
                try: print('  ', f, open(p).read().strip())
                except Exception as e: print('  ', f, 'ERR', e)
    try: print(open(c + '/pp_dpm_sclk').read())
    except Exception as e: print('ERR', e)
