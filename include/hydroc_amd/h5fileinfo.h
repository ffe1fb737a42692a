// hydroc_amd/h5fileinfo.h -- the reference's view of a BEMIO HDF5 file (include/hydroc/h5fileinfo.h:35-260: HydroData and its reader
// H5FileInfo) for programs that look at the file themselves:
//
//     HydroData infos = H5FileInfo(h5fname, 2).ReadH5Data();          // tests/h5fileinfo_t01.cpp:20
//     auto rirf_time_vector = infos.GetRIRFTimeVector();              //                      :22
//
// Same class names, getter names, argument order (0-based body first) and scaling rules; host side only -- the file is read by the
// library's HDF5 reader (hc_h5_read, no GPU involved), and TestHydro does not go through this class (it hands the file to its device
// contexts with hc_load_bemio_h5).  The reference returns Eigen vectors / matrices / tensors; Eigen is not a dependency here, so vectors
// are std::vector<double> and matrices the small row-major Matrix below (rows(), cols(), operator()(i, j) as with Eigen).
//
// Scaling, as in the reference (src/h5fileinfo.cpp): inf_added_mass x rho at read (:60-61), excitation magnitudes x rho g (:73-75),
// excitation IRF x rho g (:89-90); lin_matrix and the stored K as in the file, GetHydrostaticStiffnessVal x rho g (:310-312) and
// GetRIRFVal x rho (:318-320) per access.
#pragma once

#include <cmath>
#include <cstddef>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include "../hydrochrono_amd.h"

namespace hydroc_amd {

class H5FileInfo;

class HydroData {
  public:
    // dense row-major matrix with the two accessors the callers of the reference use on an Eigen::MatrixXd
    struct Matrix {
        Matrix() = default;
        Matrix(int r, int c) : rows_(r), cols_(c), a_(static_cast<size_t>(r) * static_cast<size_t>(c), 0.0) {}
        int rows() const { return rows_; }
        int cols() const { return cols_; }
        double& operator()(int i, int j) { return a_[static_cast<size_t>(i) * static_cast<size_t>(cols_) + static_cast<size_t>(j)]; }
        double operator()(int i, int j) const { return a_[static_cast<size_t>(i) * static_cast<size_t>(cols_) + static_cast<size_t>(j)]; }
        double* data() { return a_.data(); }
        const double* data() const { return a_.data(); }

      private:
        int rows_ = 0, cols_ = 0;
        std::vector<double> a_;
    };

    struct BodyInfo {  // :37-49
        std::string body_name;                 // "body1", ...
        int body_num = 0;                      // 0-based
        double disp_vol = 0.0;                 // displaced volume at equilibrium
        std::vector<double> rirf_time_vector;  // S lags of the radiation IRF
        double rirf_timestep = 0.0;            // rirf_time_vector[1] - rirf_time_vector[0]
        std::vector<double> cg, cb;            // centre of gravity / of buoyancy, 3 entries each
        Matrix lin_matrix;                     // 6 x 6 linear restoring stiffness, as in the file
        Matrix inf_added_mass;                 // 6 x 6N added mass at infinite frequency, x rho
        std::vector<double> rirf_matrix;       // K in file order {6, 6N, S}, as in the file (GetRIRFVal applies rho)
        int rirf_dims[3] = {0, 0, 0};
    };
    struct SimulationParameters {  // :50-55
        double rho = 0.0, g = 0.0, water_depth = 0.0;
    };
    struct RegularWaveInfo {  // :56-60
        std::vector<double> freq_list;   // simulation_parameters/w
        Matrix excitation_mag_matrix;    // 6 x nw, x rho g  (the file's {6, 1, nw} with the unit wave-direction axis dropped)
        Matrix excitation_phase_matrix;  // 6 x nw
    };
    struct IrregularWaveInfo {  // :61-72
        std::vector<double> excitation_irf_time;  // L lags
        Matrix excitation_irf_matrix;             // 6 x L, x rho g
    };

    HydroData() = default;

    // getters: body number first, 0-based (:90-187)
    Matrix GetInfAddedMassMatrix(int b) const { return body(b).inf_added_mass; }
    double GetHydrostaticStiffnessVal(int b, int i, int j) const { return body(b).lin_matrix(i, j) * sim_data_.rho * sim_data_.g; }
    Matrix GetLinMatrix(int b) const { return body(b).lin_matrix; }
    double GetRIRFVal(int b, int dof, int col, int s) const {
        const BodyInfo& q = body(b);
        if (dof < 0 || dof >= q.rirf_dims[0] || col < 0 || col >= q.rirf_dims[1] || s < 0 || s >= q.rirf_dims[2])
            throw std::out_of_range("HydroData::GetRIRFVal: index out of range");
        const size_t at = (static_cast<size_t>(dof) * static_cast<size_t>(q.rirf_dims[1]) + static_cast<size_t>(col)) * static_cast<size_t>(q.rirf_dims[2]) +
                          static_cast<size_t>(s);
        return q.rirf_matrix[at] * sim_data_.rho;
    }
    double GetDispVolVal(int b) const { return body(b).disp_vol; }
    std::vector<double> GetCGVector(int b) const { return body(b).cg; }
    std::vector<double> GetCBVector(int b) const { return body(b).cb; }
    double GetExcitationIRFVal(int b, int dof, int s) const { return irreg(b).excitation_irf_matrix(dof, s); }
    Matrix GetExcitationIRF(int b) const { return irreg(b).excitation_irf_matrix; }
    int GetRIRFDims(int i) const {
        if (i < 0 || i > 2) throw std::out_of_range("HydroData::GetRIRFDims: dimension 0, 1 or 2");
        return body(0).rirf_dims[i];
    }
    // the lags all bodies share; a body whose lags differ from body 0's by more than 1e-10 is an error (src/h5fileinfo.cpp:329-343)
    std::vector<double> GetRIRFTimeVector() const {
        const std::vector<double>& t0 = body(0).rirf_time_vector;
        for (size_t ii = 1; ii < body_data_.size(); ++ii) {
            const std::vector<double>& t = body_data_[ii].rirf_time_vector;
            if (t.size() != t0.size())
                throw std::runtime_error("RIRF time vectors have to be exactly the same for all bodies. Body " + std::to_string(ii) + " has " +
                                         std::to_string(t.size()) + " entries, body 0 has " + std::to_string(t0.size()) + ".");
            for (size_t jj = 0; jj < t.size(); ++jj)
                if (std::abs(t[jj] - t0[jj]) > 1e-10)
                    throw std::runtime_error("RIRF time vectors have to be exactly the same for all bodies. Difference found in body " +
                                             std::to_string(ii) + " at time index " + std::to_string(jj) + ".");
        }
        return t0;
    }
    double GetRhoVal() const { return sim_data_.rho; }

    // the chunks themselves (:198-225)
    std::vector<BodyInfo>& GetBodyInfos() { return body_data_; }
    SimulationParameters& GetSimulationInfo() { return sim_data_; }
    std::vector<RegularWaveInfo>& GetRegularWaveInfos() { return reg_wave_data_; }
    std::vector<IrregularWaveInfo>& GetIrregularWaveInfos() { return irreg_wave_data_; }

  private:
    friend class H5FileInfo;
    void resize(int num_bodies) {
        body_data_.resize(static_cast<size_t>(num_bodies));
        reg_wave_data_.resize(static_cast<size_t>(num_bodies));
        irreg_wave_data_.resize(static_cast<size_t>(num_bodies));
    }
    const BodyInfo& body(int b) const {
        if (b < 0 || b >= static_cast<int>(body_data_.size())) throw std::out_of_range("HydroData: body number out of range");
        return body_data_[static_cast<size_t>(b)];
    }
    const IrregularWaveInfo& irreg(int b) const {
        if (b < 0 || b >= static_cast<int>(irreg_wave_data_.size())) throw std::out_of_range("HydroData: body number out of range");
        return irreg_wave_data_[static_cast<size_t>(b)];
    }
    std::vector<BodyInfo> body_data_;
    SimulationParameters sim_data_;
    std::vector<RegularWaveInfo> reg_wave_data_;
    std::vector<IrregularWaveInfo> irreg_wave_data_;
};

class H5FileInfo {  // :230-262
  public:
    H5FileInfo(std::string file, int num_bodies) : h5_file_name_(std::move(file)), num_bodies_(num_bodies) {}
    H5FileInfo()                               = delete;
    H5FileInfo(const H5FileInfo&)              = default;
    H5FileInfo& operator=(const H5FileInfo&)   = default;
    H5FileInfo(H5FileInfo&&)                   = default;
    H5FileInfo& operator=(H5FileInfo&&)        = default;
    ~H5FileInfo()                              = default;

    // body1 .. body<num_bodies> of the file (src/h5fileinfo.cpp:27-153).  std::runtime_error when the file cannot be opened or read, a
    // dataset is missing or has the wrong shape ("Unable to open/read HDF5 hydro data file: <path>", :167-179)
    HydroData ReadH5Data() {
        hc_h5data* raw = nullptr;
        if (hc_h5_read(h5_file_name_.c_str(), num_bodies_, &raw) != HC_OK) throw std::runtime_error(hc_last_error(nullptr));
        struct Release {
            hc_h5data* p;
            ~Release() { hc_h5_free(p); }
        } release{raw};
        auto ok = [](int rc) {
            if (rc != HC_OK) throw std::runtime_error(hc_last_error(nullptr));
        };
        HydroData d;
        d.resize(num_bodies_);
        ok(hc_h5_get_sizes(raw, nullptr, &d.sim_data_.rho, &d.sim_data_.g, &d.sim_data_.water_depth, 0, nullptr, nullptr, nullptr));
        const int D = 6 * num_bodies_;
        for (int b = 0; b < num_bodies_; ++b) {
            int S = 0, nw = 0, L = 0;
            ok(hc_h5_get_sizes(raw, nullptr, nullptr, nullptr, nullptr, b, &S, &nw, &L));
            HydroData::BodyInfo& q = d.body_data_[static_cast<size_t>(b)];
            q.body_name            = "body" + std::to_string(b + 1);
            q.body_num             = b;
            q.rirf_time_vector.assign(static_cast<size_t>(S), 0.0);
            q.cg.assign(3, 0.0);
            q.cb.assign(3, 0.0);
            q.lin_matrix     = HydroData::Matrix(6, 6);
            q.inf_added_mass = HydroData::Matrix(6, D);
            ok(hc_h5_get_body(raw, b, &q.disp_vol, q.cg.data(), q.cb.data(), q.lin_matrix.data(), q.inf_added_mass.data(), q.rirf_time_vector.data()));
            q.rirf_timestep = S > 1 ? q.rirf_time_vector[1] - q.rirf_time_vector[0] : 0.0;
            q.rirf_dims[0]  = 6;
            q.rirf_dims[1]  = D;
            q.rirf_dims[2]  = S;
            q.rirf_matrix.assign(static_cast<size_t>(6) * static_cast<size_t>(D) * static_cast<size_t>(S), 0.0);
            ok(hc_h5_get_rirf(raw, b, q.rirf_matrix.data()));
            HydroData::RegularWaveInfo& r = d.reg_wave_data_[static_cast<size_t>(b)];
            r.freq_list.assign(static_cast<size_t>(nw), 0.0);
            r.excitation_mag_matrix   = HydroData::Matrix(6, nw);
            r.excitation_phase_matrix = HydroData::Matrix(6, nw);
            ok(hc_h5_get_excitation_rao(raw, b, r.freq_list.data(), r.excitation_mag_matrix.data(), r.excitation_phase_matrix.data()));
            HydroData::IrregularWaveInfo& w = d.irreg_wave_data_[static_cast<size_t>(b)];
            w.excitation_irf_time.assign(static_cast<size_t>(L), 0.0);
            w.excitation_irf_matrix = HydroData::Matrix(6, L);
            ok(hc_h5_get_excitation_irf(raw, b, w.excitation_irf_time.data(), w.excitation_irf_matrix.data()));
        }
        return d;
    }

  private:
    std::string h5_file_name_;
    int num_bodies_;
};

}  // namespace hydroc_amd
