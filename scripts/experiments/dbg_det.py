import sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), 'tests'))
import numpy as np, torch
import test_gpu_detector_uint8 as T
from deepdish_amd.pipeline import make_detector
wanted = [l for l in T._labels().values() if l and l != '???']
det = make_detector(T.MODEL, wanted_labels=wanted)
frame = T._frames()[1]
rgba = np.dstack([frame[..., ::-1], np.full(frame.shape[:2] + (1,), 255, np.uint8)])
wb, wl, ws, op = T._oracle_detect(det.ssdm.weights, frame, wanted)
out = det.ssdm.invoke_device(det.ssdm.prepare_image_device(torch.from_numpy(rgba).cuda(), 480, 640, 4))
np.set_printoptions(precision=4, suppress=True, linewidth=200)
print('op count', out[3], op[3])
for i in range(10):
    print(i, out[1][i], out[2][i], out[0][i], '|', op[1][i], op[2][i], op[0][i])
from PIL import Image
got = det.detect_image(Image.fromarray(rgba, 'RGBA'))
print('got', list(zip(got[1], got[2])))
print('want', list(zip(wl, ws)))
print(np.array(got[0])); print(np.array(wb))
