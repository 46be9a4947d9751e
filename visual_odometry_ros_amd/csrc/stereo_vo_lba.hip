// stereo_vo_lba.hip — the local bundle adjustment of a StereoVO keyframe (placeholder until the next commit)
#include "stereo_vo.hpp"
int vo_svo_local_ba(vo_svo *s, vo_svo_frame_info *info) {
  (void)s;
  (void)info;
  return VO_OK;
}
