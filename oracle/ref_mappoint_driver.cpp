// oracle/ref_mappoint_driver.cpp — TEST INFRASTRUCTURE ONLY.
//
// C-ABI wrapper around the REFERENCE's own ORB_SLAM3::MapPoint, compiled from /root/reference/src/MapPoint.cc (+ ORBmatcher.cc for
// DescriptorDistance; unmodified, read in place, never copied into this repo) over oracle/slam_shim/mappoint_world.h.  Built by
// oracle/Makefile into oracle/_ref/libref_mappoint.so.  Pins MapPoint::ComputeDistinctiveDescriptors (src/MapPoint.cc:438-529).
#include <cstdint>
#include <cstring>
#include <vector>
#include <map>
#include <mutex>
#include <set>
#define protected public    // the driver sets MapPoint::mfMaxDistance for the PredictScale probe (access specifiers do not change the layout)
#include "MapPoint.h"       // the reference header, via -I/root/reference/include
#undef protected

using namespace ORB_SLAM3;

std::set<MapPoint*> KeyFrame::GetMapPoints() { std::set<MapPoint*> s; for (MapPoint* p : mvpMapPoints) if (p && !p->isBad()) s.insert(p); return s; }   // src/KeyFrame.cc:370-385

extern "C" {

// P map points; point p owns the observations [start[p], start[p+1]).  Observation o: descriptor desc[32*o]; right_of_prev[o] != 0 means
// "the right-camera observation of the same (fisheye-rig) key frame as observation o - 1"; bad_kf[o] != 0 marks its key frame bad.
// The reference iterates std::map<KeyFrame*, tuple<int,int>> (:451), i.e. in key-frame ADDRESS order; the key frames of a point are
// allocated as one array here, so that address order = observation order.
// out_desc[32*p] = MapPoint::GetDescriptor() after ComputeDistinctiveDescriptors(); has[p] = 0 if the point ended without a descriptor.
void ref_mp_distinctive(const uint8_t* desc, const int* start, const uint8_t* right_of_prev, const uint8_t* bad_kf, int P, uint8_t* out_desc, uint8_t* has) {
    Map map;
    for (int p = 0; p < P; p++) {
        const int n = start[p + 1] - start[p];
        int nkf = 0;
        for (int o = 0; o < n; o++) nkf += !right_of_prev[start[p] + o];
        KeyFrame* kfs = new KeyFrame[nkf > 0 ? nkf : 1];
        KeyFrame ref_kf;
        MapPoint* mp = new MapPoint(Eigen::Vector3f(0, 0, 1), nkf > 0 ? &kfs[0] : &ref_kf, &map);
        int k = -1;
        for (int o = 0; o < n; o++) {
            const int g = start[p] + o;
            if (!right_of_prev[g]) {
                k++;
                kfs[k].N = 1; kfs[k].mDescriptors.create(2, 32, CV_8UC1); kfs[k].mvuRight.assign(2, -1.0f);
                memcpy(kfs[k].mDescriptors.ptr(0), desc + 32 * (size_t)g, 32);
                kfs[k].mbBadKF = bad_kf[g] != 0;
                mp->AddObservation(&kfs[k], 0);
            } else {
                kfs[k].N = 2; kfs[k].NLeft = 1; kfs[k].NRight = 1;
                memcpy(kfs[k].mDescriptors.ptr(1), desc + 32 * (size_t)g, 32);
                mp->AddObservation(&kfs[k], 1);            // idx >= NLeft: stored as the right index (src/MapPoint.cc:187-189)
            }
        }
        mp->ComputeDistinctiveDescriptors();
        const cv::Mat d = mp->GetDescriptor();
        has[p] = !d.empty();
        if (has[p]) memcpy(out_desc + 32 * (size_t)p, d.ptr(0), 32);
        delete mp;
        delete[] kfs;
    }
}

// MapPoint::PredictScale(currentDist, Frame*) of the reference's own src/MapPoint.cc:714-731 for n (max_distance, distance) pairs: pins which
// log the reference calls (std::log(float): the translation unit is `using namespace std`) for the device's k_frustum / k_project_points.
void ref_mp_predict_scale(const float* max_dist, const float* dist, int n, float log_scale_factor, int nlevels, int* out) {
    Map map;
    KeyFrame ref_kf;
    Frame F;
    F.mfLogScaleFactor = log_scale_factor; F.mnScaleLevels = nlevels;
    MapPoint* mp = new MapPoint(Eigen::Vector3f(0, 0, 1), &ref_kf, &map);
    for (int i = 0; i < n; i++) {
        mp->mfMaxDistance = max_dist[i];
        out[i] = mp->PredictScale(dist[i], &F);
    }
    delete mp;
}

}  // extern "C"
